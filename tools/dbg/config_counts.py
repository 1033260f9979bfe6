#!/usr/bin/env python3
"""Traversal counters of one frame (RT_FLAG_COUNTERS): nodes, primitive tests and packet steps per ray / packet.
usage: tools/dbg/config_counts.py C2|C5 [aa]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else "C5"
aa = int(sys.argv[2]) if len(sys.argv) > 2 else (4 if which == "C5" else 1)
with tempfile.TemporaryDirectory() as tmp:
    if which == "C5":
        flat = rt.FlatScene.from_json(scenes.c5_scene(tmp, int(os.environ.get("C5_GRID", "1581")))); W, H = 3840, 2160
    else:
        flat = rt.FlatScene.from_json(scenes.c2_scene(tmp, 224)); W, H = 1920, 1080
    sc = rt.Scene(flat, 0)
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    frame = torch.zeros(H, W, 3, device="cuda")
    st = sc.render_into(frame.data_ptr(), W, H, aa=(aa, aa), camera=cam, flags=1)
    rn, rs = st["rays_nearest"], st["rays_shadow"]
    print(which, "aa", aa, {k: st[k] for k in st if k.startswith(("nodes", "prims", "steps", "rays"))})
    print(f"nearest: {st['nodes_nearest'] / rn:.2f} nodes, {st['prims_nearest'] / rn:.2f} prims per ray, {st['steps_nearest'] / (rn / 64):.1f} steps per packet")
    print(f"shadow:  {st['nodes_shadow'] / rs:.2f} nodes, {st['prims_shadow'] / rs:.2f} prims per ray, {st['steps_shadow'] / (rs / 64):.1f} steps per 64 rays")
    sc.close()
