#!/usr/bin/env python3
"""Where the set-up of the full-size C5 scene goes: load (file, reference tree) and upload (sub-trees, treelets, quads, copies), RT_LOAD_TIMING lines."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RT_LOAD_TIMING"] = "1"
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
torch.cuda.init(); torch.zeros(1, device="cuda")
with tempfile.TemporaryDirectory() as tmp:
    p = scenes.c5_scene(tmp, int(os.environ.get("C5_GRID", "1581")))
    for i in range(2):
        t0 = time.perf_counter(); flat = rt.FlatScene.from_json(p); t1 = time.perf_counter()
        sc = rt.Scene(flat, 0); t2 = time.perf_counter()
        print(f"load {t1 - t0:.2f} s, upload {t2 - t1:.2f} s, together {t2 - t0:.2f} s", flush=True)
        sc.close(); del flat
