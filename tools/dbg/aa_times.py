#!/usr/bin/env python3
"""C2 at 1080p with the sample grids bench.py uses for N = 2, 4, 8 GPUs (each GPU's share of rays is the N = 1 frame's; here one GPU renders
the whole multi-sample frame), three frames in flight, with the row sums of multi-sample frames on and off (RT_NO_ROW_SUMS)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c2_scene(tmp, 224))
    sc = rt.Scene(flat, 0)
    W, H = 1920, 1080
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    bufs = [torch.zeros(H, W, 3, device="cuda") for _ in range(3)]
    for aa in ((1, 1), (2, 1), (2, 2), (4, 2)):
        for off in (False, True):
            if off: os.environ["RT_NO_ROW_SUMS"] = "1"
            else: os.environ.pop("RT_NO_ROW_SUMS", None)
            st = sc.render_into(bufs[0].data_ptr(), W, H, aa=aa, camera=cam)
            torch.cuda.synchronize()
            pending, n = [], 30
            t0 = time.perf_counter()
            for i in range(n):
                if len(pending) == 3: sc.render_wait(pending.pop(0))
                pending.append(sc.render_submit(bufs[i % 3].data_ptr(), W, H, aa=aa, camera=cam))
            for t in pending: sc.render_wait(t)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            rays = st["rays_nearest"] + st["rays_shadow"]
            print(f"aa {aa} row sums {'off' if off else 'on '}: {dt * 1e3:.3f} ms/frame, {rays / dt / 1e6:.0f} Mrays/s", flush=True)
    sc.close()
