#!/usr/bin/env python3
"""Throughput of whole frames rendered on P handles from P host threads (frames in flight = P); RT_LANES from the env."""
import os, sys, time, tempfile, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes

W, H = 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c2_scene(tmp, grid=224))
cam = rt.Camera(flat.fov, W, H, eye=(0.0, 1.5, 1.0), rot_x=-0.3)
for P in (1, 2, 3):
    handles = [rt.Scene(flat, 0) for _ in range(P)]
    frames = [torch.zeros(H, W, 3, device="cuda") for _ in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    rays = [0] * P
    def work(b, reps):
        for _ in range(reps):
            st = handles[b].render_into(frames[b].data_ptr(), W, H, camera=cam, stream=streams[b].cuda_stream)
            rays[b] = st["rays_nearest"] + st["rays_shadow"]
    def run(reps):
        th = [threading.Thread(target=work, args=(b, reps)) for b in range(P)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
    run(3)
    n = 12
    t0 = time.perf_counter(); run(n); dt = (time.perf_counter() - t0) / (n * P)
    print(f"lanes={os.environ.get('RT_LANES', 'default')} P={P}: {dt * 1e3:.3f} ms/frame, {rays[0] / dt / 1e6:.0f} Mrays/s", flush=True)
    for h in handles: h.close()
