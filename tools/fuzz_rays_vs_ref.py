#!/usr/bin/env python3
"""The oracle's ray queries (oracle_intersect_batch / oracle_shadow_batch) against the COMPILED REFERENCE's BVH::findNearest / BVH::calcShadow
(oracle/_ref/ref_driver ... rays; container only, no GPU) on awkward rays over random scenes: zero, denormal, tiny and huge direction components,
no direction at all, origins on box faces, lights at the shaded point itself -- the inputs of tools/fuzz_rays.py, which compares the device
with the oracle on the GPU box.  Ids and lit flags exact, t and the attenuation colour bit for bit.
usage: tools/fuzz_rays_vs_ref.py [first_seed] [count] [rays per scene]"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def reference_queries(name, cwd, e, d, p, l):
    with tempfile.TemporaryDirectory() as t:
        fin, fout = os.path.join(t, "in.bin"), os.path.join(t, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<Q", len(e))); f.write(np.hstack([e, d]).astype("<f4").tobytes())
            f.write(struct.pack("<Q", len(p))); f.write(np.hstack([p, l]).astype("<f4").tobytes())
        subprocess.run([DRIVER, name, "0", "0", "rays", fin, fout], cwd=cwd, check=True, capture_output=True)
        buf = open(fout, "rb").read()
    near = np.frombuffer(buf, np.dtype([("id", "<i4"), ("t", "<f4")]), len(e), 0)
    shad = np.frombuffer(buf, np.dtype([("lit", "<i4"), ("shadow", "<f4", 3)]), len(p), len(e) * 8)
    return near, shad


def one(seed, n):
    rng = np.random.default_rng(seed)
    with tempfile.TemporaryDirectory() as tmp:
        path = scenes.random_scene(os.path.join(tmp, "scenes"), seed=seed)
        flat = rt.FlatScene.from_json(path)
        e, d = scenes.awkward_rays(rng, n)
        o = oracle.intersect_batch(flat, e, d)
        hit = o["hit"].astype(bool)
        p = (e + o["t"][:, None] * d)[hit].astype(np.float32)          # shadow queries from the hit points of those rays
        p = p[np.isfinite(p).all(axis=1)]
        l = rng.uniform([-4, 2, -9], [4, 6, 1], (len(p), 3)).astype(np.float32)
        l[::7] = p[::7]                                                   # light at the point itself
        near, shad = reference_queries(os.path.basename(path)[:-5], tmp, e, d, p, l)
    os_ = oracle.shadow_batch(flat, p, l)
    ref_hit = near["id"] >= 0
    bad_near = int((ref_hit != hit).sum() + (near["id"][hit & ref_hit] != o["id"][hit & ref_hit]).sum() + (bits(near["t"][hit & ref_hit]) != bits(o["t"][hit & ref_hit])).sum())
    lit = shad["lit"] != 0
    bad_shad = int((lit != (os_["lit"] != 0)).sum() + (bits(shad["shadow"][lit]) != bits(os_["shadow"][lit])).any(axis=1).sum())
    return bad_near, bad_shad, int(hit.sum()), len(p), int((~np.isfinite(o["t"][hit])).sum())


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
    bad = 0
    for seed in range(first, first + count):
        bn, bs, nh, ns, nonfinite = one(seed, n)
        bad += bool(bn or bs)
        print(f"seed {seed}: {'identical' if not (bn or bs) else 'DIFFERS'}: {bn} of {n} nearest queries differ (id or t; {nh} hits, {nonfinite} with non-finite t), {bs} of {ns} shadow queries (lit or colour)", flush=True)
    print(f"{count - bad} of {count} scenes: the oracle's ray queries are bit-identical to the compiled reference's")
    sys.exit(1 if bad else 0)
