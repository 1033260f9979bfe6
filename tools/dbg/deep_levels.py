#!/usr/bin/env python3
"""Deeper levels through the packet kernels or through the per-lane walk: the C2 height field made reflective (untextured, so the
level loop runs ahead of the host) under a mirror sphere and a glass sphere, 1920x1080, depth limit 6."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes

with tempfile.TemporaryDirectory() as tmp:
    with open(scenes.c2_scene(tmp, grid=224)) as f:
        doc = json.load(f)
    mesh, s1, s2 = doc["objects"][:3]
    if os.environ.get("TEXTURED") != "1":   # (TEXTURED=1: the field stays a texturemesh, so its hits leave texel records)
        mesh["type"] = "mesh"; mesh.pop("texture"); mesh.pop("uvCoords")
        mesh["material"]["diffuse"] = [0.5, 0.6, 0.4]
    mesh["material"]["reflective"] = [0.4, 0.4, 0.4]
    for sph in (s1, s2):
        sph["type"] = "sphere"; sph.pop("bumpmap"); sph.pop("bumpdepth")
    s1["material"]["reflective"] = [0.5, 0.5, 0.5]
    s2["material"]["transmissive"] = [0.6, 0.6, 0.6]; s2["material"]["refraction"] = 1.3
    path = os.path.join(tmp, "mirror_field.json")
    with open(path, "w") as f:
        json.dump(doc, f)
    flat = rt.FlatScene.from_json(path)
    sc = rt.Scene(flat, 0)
    W, H = (int(v) for v in os.environ.get("SIZE", "1920x1080").split("x"))
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    frame = torch.zeros(H, W, 3, device="cuda")
    for levels in ("8", "1", "2", "0"):
        os.environ["RT_PACKET_LEVELS"] = levels
        sc.render_into(frame.data_ptr(), W, H, camera=cam, max_depth=6)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            st = sc.render_into(frame.data_ptr(), W, H, camera=cam, max_depth=6)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        rays = st["rays_nearest"] + st["rays_shadow"]
        print(f"RT_PACKET_LEVELS={levels}: {dt * 1e3:.3f} ms/frame, {rays / 1e6:.2f} Mrays, {rays / dt / 1e6:.0f} Mrays/s, levels {st['levels']}", flush=True)
    sc.close()
