#!/usr/bin/env python3
"""Feasibility check: render one frame as B interleaved tile shards on B streams concurrently (B handles, B host threads)."""
import os, sys, time, tempfile, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes

W, H = 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c2_scene(tmp, grid=224))
cam = rt.Camera(flat.fov, W, H, eye=(0.0, 1.5, 1.0), rot_x=-0.3)
frame = torch.zeros(H, W, 3, device="cuda")
for B in (1, 2, 3, 4, 6, 8):
    handles = [rt.Scene(flat, 0) for _ in range(B)]
    streams = [torch.cuda.Stream() for _ in range(B)]
    rays = [0] * B
    def work(b, reps):
        for _ in range(reps):
            st = handles[b].render_into(frame.data_ptr(), W, H, camera=cam, tile_rank=b, tile_ranks=B, stream=streams[b].cuda_stream)
            rays[b] = st["rays_nearest"] + st["rays_shadow"]
    def run(reps):
        th = [threading.Thread(target=work, args=(b, reps)) for b in range(B)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
    run(3)
    t0 = time.perf_counter(); run(20); dt = (time.perf_counter() - t0) / 20
    print(f"B={B}: {dt * 1e3:.3f} ms/frame, {sum(rays) / dt / 1e6:.0f} Mrays/s", flush=True)
    del handles
