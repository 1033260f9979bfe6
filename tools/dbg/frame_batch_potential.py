#!/usr/bin/env python3
"""What would several frames in one launch buy the latency-bound configurations?  A frame of k times the rows stands in for k frames of the same scene
batched into every launch: ms per 1080p-equivalent (C3) or 480-row-equivalent (bundled scenes) frame, whole frames on the lanes, 6 in flight."""
import gzip, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import scenes


def piped(sc, cam, W, H, max_depth, D=6, reps=12):
    bufs = [torch.zeros(H, W, 3, device="cuda") for _ in range(D)]
    for rep in range(2):
        pend, n = [], D * reps
        t0 = time.perf_counter()
        for i in range(n):
            if len(pend) == D:
                sc.render_wait(pend.pop(0))
            pend.append(sc.render_submit(bufs[i % D].data_ptr(), W, H, camera=cam, max_depth=max_depth, flags=rt.RT_FLAG_WHOLE_FRAME))
        for t in pend:
            sc.render_wait(t)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with tempfile.TemporaryDirectory() as tmp:
    cases = [("C3", rt.FlatScene.from_json(scenes.c3_scene(tmp)), 1920, 1080, 5, (0, 1.0, 2.0), -0.25)]
    for ch in "ahil":
        dst = os.path.join(tmp, f"{ch}.rtscene")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "scenes", f"{ch}.rtscene.gz"), "rb") as a, open(dst, "wb") as b:
            shutil.copyfileobj(a, b)
        cases.append((ch, rt.FlatScene.load(dst), 640, 480, 16, (0, 0, 0), 0.0))
    for name, flat, W, H, depth, eye, rot in cases:
        sc = rt.Scene(flat, 0)
        out = []
        for k in (1, 2, 4):
            cam = rt.Camera(flat.fov, W, H * k, eye=eye, rot_x=rot)
            out.append(piped(sc, cam, W, H * k, depth) / k * 1e3)
        print(f"{name}: ms per frame-equivalent with 1 / 2 / 4 frames' worth of rows per launch: " + " / ".join(f"{v:.4f}" for v in out), flush=True)
        sc.close()
