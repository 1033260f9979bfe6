import os, sys, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, raytracer_amd as rt
from raytracer_amd import scenes
x, y = int(sys.argv[1]), int(sys.argv[2])
W, H = 3840, 2160
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c5_scene(tmp, grid=1581))
    sc = rt.Scene(flat, 0)
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    img = sc.render(W, H, camera=cam)
    print("env", {k: v for k, v in os.environ.items() if k.startswith("RT_")}, "pixel", img[y, x], "neighbours", img[y, x - 1], img[y, x + 1])
    def primary_targets_np(cam, W, H):
        f = np.float32
        eye, facing, right, up = (np.array(v[:], f) for v in (cam.c.eye, cam.c.facing, cam.c.right, cam.c.up))
        x = np.arange(W, dtype=f)[None, :, None]
        y = np.arange(H, dtype=f)[:, None, None]
        a = f(2) * ((x + f(0.5)) / f(W) - f(0.5))
        b = f(2) * ((y + f(0.5)) / f(H) - f(0.5))
        s = ((eye + facing)[None, None, :] + right[None, None, :] * a) + up[None, None, :] * b
        return s.reshape(-1, 3).astype(f)
    s = primary_targets_np(cam, W, H).reshape(H, W, 3)[y, x][None, :]
    e = np.array(cam.c.eye[:], np.float32)[None, :]
    r = sc.trace_batch(e, s)
    print("per-lane trace:", r["hit"], r["id"], r["t"], r["rgb"])
    sc.close()
