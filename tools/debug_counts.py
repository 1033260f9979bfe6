#!/usr/bin/env python3
"""Finds the rays of a fuzz scene whose ray trees differ in size between device and oracle (tools/fuzz_parity.py reports only totals):
splits the frame's rays into blocks until single rays are left, then prints them with the device's narration of their tree.
usage: tools/debug_counts.py seed [W H objects]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import raytracer_amd as rt
from raytracer_amd import scenes

seed = int(sys.argv[1]); W = int(sys.argv[2]) if len(sys.argv) > 2 else 256; H = int(sys.argv[3]) if len(sys.argv) > 3 else 144
n_objects = int(sys.argv[4]) if len(sys.argv) > 4 else 40
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.random_scene(tmp, seed=seed, n_objects=n_objects))
sc = rt.Scene(flat, 0)
cam = rt.Camera(flat.fov, W, H, eye=(0, 1.0, 1.5), rot_x=-0.2)
s = oracle.primary_targets(cam, W, H)
e = np.tile(np.array(cam.c.eye[:], np.float32), (W * H, 1))


def counts(idx):
    sc.trace_batch(e[idx], s[idx], max_depth=6, seed=7)
    st = sc.last_stats
    o = oracle.trace_batch(flat, e[idx], s[idx], max_depth=6, seed=7, threads=16)
    return (st["rays_nearest"], st["rays_shadow"]), (o["rays_nearest"], o["rays_shadow"])


todo, found = [np.arange(W * H)], []
while todo:
    idx = todo.pop()
    d, o = counts(idx)
    if d == o:
        continue
    if len(idx) == 1:
        found.append((int(idx[0]), d, o))
        continue
    todo += [idx[: len(idx) // 2], idx[len(idx) // 2:]]
for i, d, o in found:
    print(f"ray {i} (pixel {i % W} {i // W}): device {d}, oracle {o}")
    hit, rgb, text = sc.trace_pick(e[i], s[i], max_depth=6, seed=7)
    r = oracle.trace_batch(flat, e[i:i + 1], s[i:i + 1], max_depth=6, seed=7)
    print("device rgb", rgb, "oracle rgb", r["rgb"][0], "failed exit searches in the oracle's tree:", oracle.state_cases())
    print(text)
print(len(found), "rays differ")
