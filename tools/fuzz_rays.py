#!/usr/bin/env python3
"""Fuzzing rt_intersect_batch / rt_shadow_batch against the oracle with awkward rays on random scenes: directions with
zero components (1/d infinite: the slab test's 0 * inf cases), denormal components, origins inside boxes and on
surfaces, zero-length and huge directions, rays along box faces.
usage: tools/fuzz_rays.py [first_seed] [count]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import raytracer_amd as rt
from raytracer_amd import scenes


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def one(seed, n=60000):
    with tempfile.TemporaryDirectory() as tmp:
        flat = rt.FlatScene.from_json(scenes.random_scene(tmp, seed=seed))
    sc = rt.Scene(flat, 0)
    rng = np.random.default_rng(seed)
    e, d = scenes.awkward_rays(rng, n)
    r = sc.intersect_batch(e, d)
    o = oracle.intersect_batch(flat, e, d)
    ok_i = np.array_equal(r["hit"], o["hit"]) and np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    # shadow queries from hit points of those rays towards random light positions
    hit = o["hit"].astype(bool)
    p = (e + o["t"][:, None] * d)[hit].astype(np.float32)
    l = rng.uniform([-4, 2, -9], [4, 6, 1], (len(p), 3)).astype(np.float32)
    l[::7] = p[::7]                                                         # light at the point itself
    rs = sc.shadow_batch(p, l)
    os_ = oracle.shadow_batch(flat, p, l)
    same_lit = np.array_equal(rs["lit"], os_["lit"])
    dsh = float(np.abs(rs["shadow"][os_["lit"].astype(bool)] - os_["shadow"][os_["lit"].astype(bool)]).max()) if os_["lit"].any() else 0.0
    sc.close()
    return ok_i, same_lit, dsh, int(hit.sum()), int((~np.isfinite(r["t"])).sum())


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    bad = 0
    for seed in range(first, first + count):
        ok_i, same_lit, dsh, nh, nonfinite = one(seed)
        ok = ok_i and same_lit and dsh < 1e-6
        bad += not ok
        print(f"seed {seed}: {'ok ' if ok else 'BAD'} nearest (hit, id, t bitwise) {ok_i}; shadow lit {same_lit}, attenuation max diff {dsh:.3g} ({nh} hits, {nonfinite} non-finite t)", flush=True)
    print(f"{count - bad} of {count} seeds agree with the oracle")
    sys.exit(1 if bad else 0)
