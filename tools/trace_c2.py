#!/usr/bin/env python3
"""One C2 bench frame on the instrumented library (make prof; RT_LIB_PATH), single lane, with the step trace dumped: feeds tools/trace_summary.py."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c2_scene(tmp, grid=224))
sc = rt.Scene(flat, 0)
W, H = 1920, 1080
cam = rt.Camera(flat.fov, W, H, eye=(0.0, 1.5, 1.0), rot_x=-0.3)
buf = torch.zeros(H, W, 3, device="cuda")
for _ in range(2):
    st = sc.render_into(buf.data_ptr(), W, H, camera=cam, flags=rt.RT_FLAG_COUNTERS)
print(st)
sc.close()
