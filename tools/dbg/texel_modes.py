#!/usr/bin/env python3
"""The mirror-over-terrain scene of test_texel_state_with_many_records rendered by the level loop that runs ahead, by the loop that
reads the counters back (flags=2 or RT_NO_LEVELS_AHEAD=1), and compared: tools/dbg/texel_modes.py out.npy [flags]"""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd.scenes import _mesh_json, height_field, tiles_texture, write_bmp24
out = sys.argv[1]; flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with tempfile.TemporaryDirectory() as tmp:
    write_bmp24(os.path.join(tmp, "t.bmp"), tiles_texture(64, 8))
    tris, uvs = height_field(6, seed=3, x0=-6.0, x1=6.0, z0=-14.0, z1=-1.0)
    mesh = {"type": "texturemesh", "texture": "t.bmp", "material": {"specular": [0.3, 0.3, 0.3], "shininess": 10.0, "reflective": [0.5, 0.4, 0.6]}}
    mesh.update(_mesh_json(tris, uvs))
    scene = {"camera": {"field": 60, "background": [0.1, 0.1, 0.2]},
             "objects": [mesh, {"type": "box", "point1": [-8, 1.5, -16], "point2": [8, 2.0, 1], "material": {"diffuse": [0.1, 0.1, 0.1], "reflective": [0.8, 0.8, 0.8]}},
                         {"type": "sphere", "position": [0.5, 0.2, -6], "radius": 0.7, "material": {"diffuse": [0.5, 0.2, 0.2], "transmissive": [0.6, 0.6, 0.6], "refraction": 1.3}}],
             "lights": [{"type": "ambient", "color": [0.3, 0.3, 0.3]}, {"type": "point", "color": [0.8, 0.8, 0.7], "position": [0, 1.2, -5]},
                        {"type": "directional", "color": [0.3, 0.3, 0.4], "direction": [0.2, -1, -0.3]},
                        {"type": "rectangular", "color": [0.4, 0.4, 0.3], "position": [1.0, 1.4, -7.0], "normal": [0, -1, 0], "orientation": [1, 0, 0],
                         "width": 1.0, "height": 0.6, "samples": 3}]}
    path = os.path.join(tmp, "mirror_terrain.json")
    open(path, "w").write(json.dumps(scene))
    flat = rt.FlatScene.from_json(path)
    sc = rt.Scene(flat, 0)
    W, H = 640, 360
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.0, -7.0), rot_x=-1.3)
    img = sc.render(W, H, camera=cam, max_depth=5, seed=9, flags=flags)
    print(out, sc.last_stats["rays_nearest"], sc.last_stats["rays_shadow"], sc.last_stats["levels"], float(img.sum()))
    np.save(out, img)
    sc.close()
