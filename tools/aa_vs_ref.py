#!/usr/bin/env python3
"""Supersampled frames (the reference's antialias mode, q1.cpp:131-141: four samples per pixel, averaged) of random scenes from a moved and turned
camera: oracle_render against the compiled reference (oracle/_ref/ref_driver with aa_grid = 2; container only, no GPU).  Float RGB bit for bit
and both ray counts; area lights on the reference's own rand() stream.   usage: tools/aa_vs_ref.py [first_seed] [count]"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from refdump import read_ref_dump  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    W, H, cam, bad = 64, 40, (0.2, 1.0, 1.5, -0.2, 0.1), 0
    for seed in range(first, first + count):
        with tempfile.TemporaryDirectory() as tmp:
            path = scenes.random_scene(os.path.join(tmp, "scenes"), seed=seed, n_objects=30)
            out = os.path.join(tmp, "o.bin")
            subprocess.run([DRIVER, os.path.basename(path)[:-5], str(W), str(H), out, "0", "2"] + [repr(float(v)) for v in cam], cwd=tmp, check=True, capture_output=True)
            g = read_ref_dump(out)
            flat = rt.FlatScene.from_json(path)
        c = rt.Camera(flat.fov, W, H, eye=cam[:3], rot_x=cam[3], rot_y=cam[4])
        ctypes.CDLL(None).srand(1)
        rgb, nn, ns = oracle.render(flat, c, W, H, aa=2, seed=0)
        same = np.array_equal(rgb.reshape(-1, 3).view(np.uint32), g["rgb"].view(np.uint32)) and nn == g["rays_nearest"] and ns == g["rays_shadow"]
        bad += not same
        print(f"seed {seed}: {'identical' if same else 'DIFFERS'} ({nn} nearest, {ns} shadow rays)", flush=True)
    print(f"{count - bad} of {count} supersampled frames bit-identical to the compiled reference")
    sys.exit(1 if bad else 0)
