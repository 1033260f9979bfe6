#!/usr/bin/env python3
"""tests/golden/ref_cameras.npz: what the compiled reference's setFacing() and s(x, y) (q1.cpp:30-53) give for twenty cameras -- eye positions,
rotations about both axes, near-zero angles that q1.cpp snaps to 0 -- on 16x9 and 7x13 frames (oracle/_ref/ref_driver, container only).
The driver needs some scene to load; the cameras do not depend on it apart from the field of view (60 degrees)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from fuzz_oracle_vs_ref import drive  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    cams = [(0, 0, 0, 0, 0), (1, 2, 3, 0.5, 0), (1, 2, 3, 0, 0.7), (-1, 0.5, 2, -0.3, 2.5), (0, 0, 0, 1.5, -3.0), (0, 1, 0, -1.2, 0.3),
            (0, 0, 0, 1e-6, 1e-6), (0, 0, 0, 3.14159, 0.1)] + [tuple(float(v) for v in rng.uniform(-3, 3, 5)) for _ in range(12)]
    out = {"cameras": np.array(cams, np.float64), "fov": 60.0}
    with tempfile.TemporaryDirectory() as tmp:
        path = scenes.random_scene(os.path.join(tmp, "scenes"), seed=1, n_objects=3)
        for k, c in enumerate(cams):
            for W, H in ((16, 9), (7, 13)):
                g = drive(os.path.basename(path)[:-5], W, H, tmp, camera=c)
                assert g["fov"] == 60.0
                for f in ("eye", "facing", "right", "up", "s"):
                    out[f"{f}_{k}_{W}x{H}"] = g[f]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_cameras.npz"), **out)
    print(len(cams), "cameras")
