#!/usr/bin/env python3
"""tests/golden/ref_random_<seed>.npz + tests/golden/scenes/random_<seed>.rtscene.gz: the COMPILED REFERENCE's output (oracle/_ref/ref_driver,
`make -C oracle ref`; container only) on scenes of raytracer_amd.scenes.random_scene -- every primitive, material and light kind at once,
which the thirteen bundled scenes do not offer -- at 96x64 from the default camera, in the format of tools/make_golden.py.  The reference loads
the generated JSON from a temporary directory (its scene path is the relative literal "scenes/", raytracer.cpp:22).
usage: tools/make_random_golden.py seed [seed ...]        (seeds whose scene has no area light: those draw from rand())
       tools/make_random_golden.py c2_<grid>             -> ref_c2_<grid>.npz: the C2 generator (textured height field of grid*grid*2 triangles, two
                                                            bump-mapped spheres) at 160x90 from bench.py's camera; c2_224 = the full-size
                                                            benchmark scene at 256x144, outputs only
       tools/make_random_golden.py c3                    -> ref_c3.npz: BASELINE config 3 at 240x135, outputs only
       tools/make_random_golden.py c5_300 | mesh_3       -> a C5-style mesh of 180 000 triangles / the scene of meshes asked for hits on their own, 192x108, outputs only
       tools/make_random_golden.py rays_<seed>           -> ref_rays_random_<seed>.npz: BVH::findNearest / BVH::calcShadow of the reference for 20 000 awkward rays
                                                            (scenes.awkward_rays) and shadow queries from their hit points, on the random scene of that seed"""
import ctypes as C
import gzip
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_golden import GOLD, run_driver  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

if __name__ == "__main__":
    lib = C.CDLL(os.path.join(ROOT, "raytracer_amd", "lib", "libraytracer_amd.so"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    for name in [a for a in sys.argv[1:] if a.startswith("c2_")]:
        from fuzz_oracle_vs_ref import drive
        with tempfile.TemporaryDirectory() as tmp:
            path = scenes.c2_scene(os.path.join(tmp, "scenes"), grid=int(name[3:]))
            big = int(name[3:]) > 64     # the full-size scene is not stored: the tests generate it again (the generator is deterministic)
            W, H = (256, 144) if big else (160, 90)
            r = drive(os.path.basename(path)[:-5], W, H, tmp, camera=(0.0, 1.5, 1.0, -0.3, 0.0))
            np.savez_compressed(os.path.join(GOLD, f"ref_{name}.npz"), **r)
            if big:
                print(f"{name}: {W}x{H}, {r['rays_nearest']} nearest, {r['rays_shadow']} shadow rays", flush=True)
                continue
            fs = C.c_void_p()
            assert lib.rt_load_scene_json(path.encode(), None, C.byref(fs)) == 0
            flat = os.path.join(tmp, "flat.rtscene")
            assert lib.rt_flat_scene_save(fs, flat.encode()) == 0
            lib.rt_flat_scene_free(fs)
            with open(flat, "rb") as f, gzip.GzipFile(os.path.join(GOLD, "scenes", f"{name}.rtscene.gz"), "wb", mtime=0) as g:
                shutil.copyfileobj(f, g)
        print(f"{name}: {W}x{H}, {r['rays_nearest']} nearest, {r['rays_shadow']} shadow rays", flush=True)
    if "c3" in sys.argv[1:]:     # BASELINE config 3 (64 nested CSG objects), default camera, the reference's own depth limit of 16; outputs only
        from fuzz_oracle_vs_ref import drive
        with tempfile.TemporaryDirectory() as tmp:
            path = scenes.c3_scene(os.path.join(tmp, "scenes"))
            r = drive(os.path.basename(path)[:-5], 240, 135, tmp)
            np.savez_compressed(os.path.join(GOLD, "ref_c3.npz"), **r)
        print(f"c3: 240x135, {r['rays_nearest']} nearest, {r['rays_shadow']} shadow rays", flush=True)
    if "c5_300" in sys.argv[1:]:     # a mesh of 180 000 triangles: the reference's depth cap of 16 leaves it in leaves of up to 3 triangles ... 131 072 leaves
        from fuzz_oracle_vs_ref import drive
        with tempfile.TemporaryDirectory() as tmp:
            path = scenes.c5_scene(os.path.join(tmp, "scenes"), grid=300, binary=False)
            r = drive(os.path.basename(path)[:-5], 192, 108, tmp, camera=(0.0, 1.5, 1.0, -0.3, 0.0))
            np.savez_compressed(os.path.join(GOLD, "ref_c5_300.npz"), **r)
        print(f"c5_300: 192x108, {r['rays_nearest']} nearest, {r['rays_shadow']} shadow rays", flush=True)
    if "mesh_3" in sys.argv[1:]:     # meshes asked for hits on their own (Mesh::rayhit): a refractive mesh, a mesh as a CSG operand
        from fuzz_oracle_vs_ref import drive
        with tempfile.TemporaryDirectory() as tmp:
            path = scenes.mesh_scene(os.path.join(tmp, "scenes"), subdiv=3)
            r = drive(os.path.basename(path)[:-5], 192, 108, tmp)
            np.savez_compressed(os.path.join(GOLD, "ref_mesh_3.npz"), **r)
        print(f"mesh_3: 192x108, {r['rays_nearest']} nearest, {r['rays_shadow']} shadow rays", flush=True)
    for name in [a for a in sys.argv[1:] if a.startswith("rays_")]:   # rays_<seed>: awkward rays on the random scene of that seed (which must have its fixture)
        from fuzz_rays_vs_ref import reference_queries
        import oracle
        import raytracer_amd as rt
        seed = int(name[5:])
        rng = np.random.default_rng(seed)
        with tempfile.TemporaryDirectory() as tmp:
            path = scenes.random_scene(os.path.join(tmp, "scenes"), seed=seed, n_objects=40)
            e, d = scenes.awkward_rays(rng, 20000)
            o = oracle.intersect_batch(rt.FlatScene.from_json(path), e, d)      # only to place the shadow queries on surfaces
            hit = o["hit"].astype(bool)
            p = (e + o["t"][:, None] * d)[hit].astype(np.float32)
            p = p[np.isfinite(p).all(axis=1)]
            l = rng.uniform([-4, 2, -9], [4, 6, 1], (len(p), 3)).astype(np.float32)
            l[::7] = p[::7]
            near, shad = reference_queries(os.path.basename(path)[:-5], tmp, e, d, p, l)
        np.savez_compressed(os.path.join(GOLD, f"ref_rays_random_{seed}.npz"), e=e, d=d, id=near["id"], t=near["t"], p=p, l=l, lit=shad["lit"].astype(np.uint8), shadow=shad["shadow"])
        print(f"{name}: {len(e)} nearest queries ({int((near['id'] >= 0).sum())} hits), {len(p)} shadow queries", flush=True)
    for seed in (int(a) for a in sys.argv[1:] if a.isdigit()):
        with tempfile.TemporaryDirectory() as tmp:
            path = scenes.random_scene(os.path.join(tmp, "scenes"), seed=seed, n_objects=40)
            kinds = {l["type"] for l in json.load(open(path))["lights"]}
            assert not kinds & {"rectangular", "circular"}, f"seed {seed} has an area light (rand()-driven): pick another"
            r = run_driver(f"random_{seed}", 96, 64, 4096, cwd=tmp)
            np.savez_compressed(os.path.join(GOLD, f"ref_random_{seed}.npz"), **r)
            fs = C.c_void_p()
            assert lib.rt_load_scene_json(path.encode(), None, C.byref(fs)) == 0
            flat = os.path.join(tmp, "flat.rtscene")
            assert lib.rt_flat_scene_save(fs, flat.encode()) == 0
            lib.rt_flat_scene_free(fs)
            with open(flat, "rb") as f, gzip.GzipFile(os.path.join(GOLD, "scenes", f"random_{seed}.rtscene.gz"), "wb", mtime=0) as g:
                shutil.copyfileobj(f, g)
        print(f"random_{seed}: 96x64, {r['rays_nearest']} nearest, {r['rays_shadow']} shadow rays", flush=True)
