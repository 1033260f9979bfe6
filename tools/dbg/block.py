import os, sys, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, raytracer_amd as rt
from raytracer_amd import scenes
x, y = int(sys.argv[1]), int(sys.argv[2])
W, H = 3840, 2160
def targets(cam, W, H):
    f = np.float32
    eye, facing, right, up = (np.array(v[:], f) for v in (cam.c.eye, cam.c.facing, cam.c.right, cam.c.up))
    xx = np.arange(W, dtype=f)[None, :, None]; yy = np.arange(H, dtype=f)[:, None, None]
    a = f(2) * ((xx + f(0.5)) / f(W) - f(0.5)); b = f(2) * ((yy + f(0.5)) / f(H) - f(0.5))
    return (((eye + facing)[None, None, :] + right[None, None, :] * a) + up[None, None, :] * b).astype(f)
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c5_scene(tmp, grid=1581))
    sc = rt.Scene(flat, 0)
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    T = targets(cam, W, H)
    x0, y0 = x // 8 * 8, y // 8 * 8
    blk = T[y0:y0 + 8, x0:x0 + 8].reshape(-1, 3)
    lane = (y - y0) * 8 + (x - x0)
    e = np.tile(np.array(cam.c.eye[:], np.float32), (64, 1))
    for name, sel in (("block of 64", np.arange(64)), ("the ray alone", np.array([lane])), ("first half", np.arange(32) if lane < 32 else np.arange(32, 64))):
        r = sc.trace_batch(e[sel], blk[sel])
        k = int(np.nonzero(sel == lane)[0][0])
        print(name, "id", r["id"][k], "t", r["t"][k], "rgb", r["rgb"][k])
    sc.close()
