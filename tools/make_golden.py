#!/usr/bin/env python3
"""Generates the committed parity fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference and oracle/_ref/ref_driver, `make -C oracle ref`).
For every bundled scene it stores
  tests/golden/ref_<s>.npz          what oracle/_ref computed at 160x120: primary targets s, hit mask, object id,
                                    t, float RGB, per-frame ray counts, ~4k sampled secondary rays (e, d -> id, t)
                                    and ~4k sampled shadow queries (p, light -> lit, attenuation)
  tests/golden/ref_<s>_640.npz      640x480: hit mask (packed), object id (int16), RGB8 (for a subset of scenes)
  tests/golden/scenes/<s>.rtscene.gz  the scene as this repo's loader flattens it (the reference's scene files and
                                    textures are data; they are stored in the flat binary form, not as copies)
Scene m (rand()-driven area lights) additionally gets ref_m_hi.npz: the reference run on a copy of the scene
with 2048 samples per light, the "high-spp reference" for the statistical test.
"""
import ctypes as C
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from refdump import read_ref_dump  # noqa: E402

REF_SRC = "/root/reference/src"
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = "abcdefghijklm"
BIG = "acgijkl"


def run_driver(scene, W, H, cap, cwd=REF_SRC, aa=1):
    out = tempfile.mktemp(suffix=".bin")
    subprocess.run([DRIVER, scene, str(W), str(H), out, str(cap), str(aa)], cwd=cwd, check=True, capture_output=True)
    r = read_ref_dump(out)
    os.remove(out)
    return r


def main():
    os.makedirs(os.path.join(GOLD, "scenes"), exist_ok=True)
    lib = C.CDLL(os.path.join(ROOT, "raytracer_amd", "lib", "libraytracer_amd.so"))
    for s in SCENES:
        r = run_driver(s, 160, 120, 4096)
        np.savez_compressed(os.path.join(GOLD, f"ref_{s}.npz"), **r)
        fs = C.c_void_p()
        rc = lib.rt_load_scene_json(f"{REF_SRC}/scenes/{s}.json".encode(), f"{REF_SRC}/scenes/".encode(), C.byref(fs))
        assert rc == 0
        tmp = tempfile.mktemp(suffix=".rtscene")
        assert lib.rt_flat_scene_save(fs, tmp.encode()) == 0
        lib.rt_flat_scene_free(fs)
        with open(tmp, "rb") as f, gzip.GzipFile(os.path.join(GOLD, "scenes", f"{s}.rtscene.gz"), "wb", mtime=0) as g:
            shutil.copyfileobj(f, g)
        os.remove(tmp)
        print(s, "160x120", r["rays_nearest"], r["rays_shadow"], flush=True)
    for s in BIG:
        r = run_driver(s, 640, 480, 0)
        q = np.clip(np.nan_to_num(r["rgb"], nan=0.0), 0, 1)
        rgb8 = (q * np.float32(255) + np.float32(0.5)).astype(np.uint8)
        np.savez_compressed(os.path.join(GOLD, f"ref_{s}_640.npz"), W=640, H=480, fov=r["fov"], hit=np.packbits(r["hit"]),
                            id=r["id"].astype(np.int16), rgb8=rgb8, rays_nearest=r["rays_nearest"], rays_shadow=r["rays_shadow"])
        print(s, "640x480", flush=True)
    # 2x2 supersampling (the reference's antialias mode, q1.cpp:131-141) on one scene
    r = run_driver("c", 160, 120, 0, aa=2)
    np.savez_compressed(os.path.join(GOLD, "ref_c_aa2.npz"), W=160, H=120, aa=2, fov=r["fov"], rgb=r["rgb"],
                        rays_nearest=r["rays_nearest"], rays_shadow=r["rays_shadow"])
    # scene m at high sample count: run the reference on a modified COPY of the scene in a scratch directory
    tmpd = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmpd, "scenes"))
    sc = json.load(open(f"{REF_SRC}/scenes/m.json"))
    for L in sc["lights"]:
        if "samples" in L:
            L["samples"] = 2048
    json.dump(sc, open(os.path.join(tmpd, "scenes", "m.json"), "w"))
    for f in os.listdir(f"{REF_SRC}/scenes"):
        if f.endswith(".bmp"):
            os.symlink(f"{REF_SRC}/scenes/{f}", os.path.join(tmpd, "scenes", f))
    r = run_driver("m", 160, 120, 0, cwd=tmpd)
    shutil.rmtree(tmpd)
    np.savez_compressed(os.path.join(GOLD, "ref_m_hi.npz"), W=160, H=120, samples=2048, rgb=r["rgb"], hit=r["hit"], id=r["id"])
    print("m_hi done")


if __name__ == "__main__":
    main()
