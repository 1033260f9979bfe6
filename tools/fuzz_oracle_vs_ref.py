#!/usr/bin/env python3
"""Pins the oracle restatement (oracle/restate.cpp) against the COMPILED REFERENCE (oracle/_ref/ref_driver) on random scenes, beyond
the thirteen scenes the reference bundles.  Container only (needs /root/reference built by `make -C oracle ref`); no GPU.

For every seed: raytracer_amd.scenes.random_scene writes scenes/random_<seed>.json (+ its two textures) into a temporary directory,
the reference's choose_scene() loads it from there (its scene path is the relative literal "scenes/", raytracer.cpp:22) and traces
a WxH frame from the default camera; the restatement traces the same rays over this repo's loader's flat scene, drawing from libc
rand() in the reference's order (seed 0, one thread) so that area lights compare bit for bit too.  Compared: hit mask, object id, t
and float RGB bit for bit, the frame's BVH::findNearest / BVH::calcShadow call counts, and the sampled secondary rays and shadow
queries the driver intercepted.

usage: tools/fuzz_oracle_vs_ref.py [first_seed] [count] [objects per scene] [W] [H] [mixed | csg | <scale factor> | c2:<grid> | c3 | c4:<samples> | c5:<grid> | mesh:<subdivisions>]
(c2 / c3 / c4: the generators of the BASELINE configurations; the seed moves the camera)"""
import ctypes
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle  # noqa: E402
import raytracer_amd as rt  # noqa: E402
import subprocess  # noqa: E402
from make_golden import DRIVER  # noqa: E402
from refdump import read_ref_dump  # noqa: E402
from raytracer_amd import scenes  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def drive(name, W, H, cwd, camera=None):
    """oracle/_ref/ref_driver on scenes/<name>.json below cwd, from the default camera or from (eye x, y, z, rotX, rotY)."""
    out = tempfile.mktemp(suffix=".bin")
    args = [DRIVER, name, str(W), str(H), out, "4096", "1"] + ([repr(float(v)) for v in camera] if camera else [])
    subprocess.run(args, cwd=cwd, check=True, capture_output=True)
    r = read_ref_dump(out)
    os.remove(out)
    return r


def one(seed, n_objects, W, H, kind="mixed"):
    camera = None
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "scenes")
        if kind == "csg":       # random operator trees of up to 16 leaves (multi-interval lists, empty operands)
            path = scenes.random_csg_scene(d, seed=seed, n_objects=n_objects)
        elif kind.startswith("c2"):   # the BASELINE workloads themselves, at sizes the reference finishes in seconds: `seed` varies the camera
            path = scenes.c2_scene(d, grid=int(kind[3:] or 24))
            camera = (0.3 * seed, 1.5, 1.0 - 0.2 * seed, -0.3, 0.05 * seed)
        elif kind.startswith("c5"):   # a mesh the reference's depth cap of 16 leaves in fat leaves (above 131 072 triangles), as JSON
            path = scenes.c5_scene(d, grid=int(kind[3:] or 300), binary=False)
            camera = (0.3 * seed, 1.5, 1.0 - 0.2 * seed, -0.3, 0.05 * seed)
        elif kind.startswith("mesh"):   # meshes asked for hits on their own (Mesh::rayhit): a refractive mesh, a mesh as a CSG operand
            path = scenes.mesh_scene(d, subdiv=int(kind[5:] or 2))
            camera = (0.2 * seed, 0.1 * seed, 0.0, 0.0, 0.03 * seed)
        elif kind == "c3":
            path = scenes.c3_scene(d)
            camera = (0.2 * seed, 1.5 + 0.1 * seed, 2.0, -0.35, 0.04 * seed)
        elif kind.startswith("c4"):
            path = scenes.c4_scene(d, samples=int(kind[3:] or 4))
            camera = (0.1 * seed, 0.0, 0.0, 0.0, 0.02 * seed)
        else:
            path = scenes.random_scene(d, seed=seed, n_objects=n_objects)
            if kind != "mixed":     # every length scaled: the reference's absolute epsilons against tiny / huge geometry
                path = scenes.scale_scene(path, float(kind))
        g = drive(os.path.basename(path)[:-5], W, H, tmp, camera)
        flat = rt.FlatScene.from_json(path)
    n = W * H
    ctypes.CDLL(None).srand(1)
    cases = oracle.lib().oracle_state_cases
    cases.restype = ctypes.c_uint64
    cases(1)
    oracle.texel_cases()
    eyes = np.tile(g["eye"].astype(np.float32), (n, 1))
    r = oracle.trace_batch(flat, eyes, g["s"], seed=0, threads=1)
    frame_cases = int(cases(1))
    texel_cases = oracle.texel_cases()
    # pixels that differ in anything: did their ray tree run into an exit search that found no exit?  (There the reference goes on
    # from hit-cache state an EARLIER ray left, objects.cpp:123-125, and the restatement from a fixed fallback.)
    differing = np.where((r["hit"] != g["hit"]) | (r["id"] != g["id"]) | (bits(r["t"]) != bits(g["t"])) | (bits(r["rgb"]) != bits(g["rgb"])).any(axis=1))[0]
    unexplained = 0
    for i in differing:
        oracle.trace_batch(flat, eyes[:1], g["s"][i:i + 1], seed=0, threads=1)
        unexplained += int(cases(1)) == 0
    res = {
        "differing": len(differing), "unexplained": unexplained, "frame_cases": frame_cases, "texel_cases": texel_cases,
        "hit": int((r["hit"] != g["hit"]).sum()), "id": int((r["id"] != g["id"]).sum()), "t": int((bits(r["t"]) != bits(g["t"])).sum()),
        "rgb": int((bits(r["rgb"]) != bits(g["rgb"])).any(axis=1).sum()),
        "rays": (r["rays_nearest"] - int(g["rays_nearest"]), r["rays_shadow"] - int(g["rays_shadow"])),
        "sec": 0, "shd": 0, "lights": int(flat.info.n_lights), "nearest": int(g["rays_nearest"]), "shadow": int(g["rays_shadow"]),
    }
    if len(g["sec_id"]):
        a = oracle.intersect_batch(flat, g["sec_e"], g["sec_d"])
        res["sec"] = int(((a["id"] != g["sec_id"]) | (bits(a["t"]) != bits(g["sec_t"]))).sum())
    if len(g["shd_lit"]):
        a = oracle.shadow_batch(flat, g["shd_p"], g["shd_l"])
        lit = g["shd_lit"] != 0
        res["shd"] = int((a["lit"] != g["shd_lit"].astype(np.uint8)).sum() + (bits(a["shadow"][lit]) != bits(g["shd_shadow"][lit])).any(axis=1).sum())
    return res


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n_objects = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 96
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 64
    kind = sys.argv[6] if len(sys.argv) > 6 else "mixed"
    exact = explained = pixels = 0
    for seed in range(first, first + count):
        r = one(seed, n_objects, W, H, kind)
        same = not (r["differing"] or r["sec"] or r["shd"] or r["rays"] != (0, 0))
        # ray counts may move with a hidden-state pixel (its subtree differs); sampled rays are stateless queries and must agree
        ok = same or (r["unexplained"] == 0 and r["sec"] == 0 and r["shd"] == 0 and r["differing"] > 0)
        exact += same
        explained += ok
        pixels += r["differing"]
        print(f"seed {seed}: {'identical' if same else 'hidden state' if ok else 'DIFFERS'}: {r['differing']} of {W * H} pixels differ (hit {r['hit']} id {r['id']} t {r['t']} rgb {r['rgb']}), "
              f"{r['unexplained']} of them without a failed exit search in their ray tree ({r['frame_cases']} failed exit searches in the frame, {r['texel_cases']} hits lit with a later hit's texel); ray counts {r['rays'][0]:+d} {r['rays'][1]:+d}; "
              f"sampled secondary rays {r['sec']}, shadow queries {r['shd']}  ({r['lights']} lights, {r['nearest']} nearest, {r['shadow']} shadow rays)", flush=True)
    print(f"{count} random scenes ({kind}): {exact} bit-identical to the compiled reference, {explained - exact} differ only in pixels whose ray tree holds a failed exit search "
          f"({pixels} pixels of {count * W * H} in all), {count - explained} differ otherwise")
    sys.exit(0 if explained == count else 1)
