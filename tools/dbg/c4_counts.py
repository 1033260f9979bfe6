import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, raytracer_amd as rt
from raytracer_amd import scenes
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c4_scene(tmp, 64))
    sc = rt.Scene(flat, 0)
    W, H = 1920, 1080
    cam = rt.Camera(flat.fov, W, H)
    frame = torch.zeros(H, W, 3, device="cuda")
    st = sc.render_into(frame.data_ptr(), W, H, camera=cam, flags=3)
    print({k: st[k] for k in ("rays_nearest", "rays_shadow", "nodes_nearest", "prims_nearest", "nodes_shadow", "prims_shadow", "steps_nearest", "steps_shadow", "nearest_ms", "shadow_ms")})
    sc.close()
