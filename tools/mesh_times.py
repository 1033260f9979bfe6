#!/usr/bin/env python3
"""Frame time of the mesh scene (refractive icosphere + CSG mesh operand) with and without the per-mesh trees."""
import os, sys, time, tempfile, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    import raytracer_amd as rt
    from raytracer_amd import scenes
    with tempfile.TemporaryDirectory() as tmp:
        flat = rt.FlatScene.from_json(scenes.mesh_scene(tmp, int(sys.argv[1])))
    sc = rt.Scene(flat, 0)
    W, H = 1920, 1080
    frame = torch.zeros(H, W, 3, device="cuda")
    sc.render_into(frame.data_ptr(), W, H)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        st = sc.render_into(frame.data_ptr(), W, H)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"subdiv {sys.argv[1]} ({flat.info.n_mesh_tris} mesh triangles) trees={'off' if os.environ.get('RT_NO_MESH_TREES') else 'on'}: {dt * 1e3:.2f} ms/frame, levels {st['levels']}")
else:
    for sub in ("3", "5"):
        for env in ({}, {"RT_NO_MESH_TREES": "1"}):
            subprocess.run([sys.executable, __file__, sub], env=dict(os.environ, **env), check=True)
