import os, sys, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, raytracer_amd as rt
from raytracer_amd import scenes
out = sys.argv[1]
grid = int(sys.argv[2]); W, H, aa = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c5_scene(tmp, grid=grid))
    sc = rt.Scene(flat, 0)
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    img = sc.render(W, H, aa=(aa, aa), camera=cam)
    np.save(out, img)
    print(out, sc.last_stats["rays_nearest"], sc.last_stats["rays_shadow"])
    sc.close()
