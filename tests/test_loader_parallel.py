"""The loader builds the reference's tree (bvh.cpp:37-70) on several threads: triangles made in parallel, every node's std::sort run by
std_sort_par.h, subtrees split and flattened side by side.  The flat scene must come out byte for byte as the single-threaded form
with libstdc++'s own std::sort (RT_LOADER_STD_SORT=1) writes it -- on a grid mesh, whose centroids tie by the row, large enough
(217 800 triangles) for every parallel path to run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import sys
sys.path.insert(0, %r)
import raytracer_amd as rt
from raytracer_amd import scenes
flat = rt.FlatScene.from_json(scenes.c5_scene(sys.argv[1], 330, binary=True))
assert flat.info.n_tris == 217800, flat.info.n_tris
flat.save(sys.argv[2])
""" % ROOT


def test_parallel_build_equals_the_sequential_std_sort_build(tmp_path):
    out = {}
    for name, extra in (("parallel", {}), ("plain", {"RT_LOADER_STD_SORT": "1"})):
        path = str(tmp_path / f"{name}.rtscene")
        subprocess.run([sys.executable, "-c", SCRIPT, str(tmp_path / name), path], check=True, env={**os.environ, **extra}, timeout=300)
        with open(path, "rb") as f:
            out[name] = f.read()
    assert len(out["parallel"]) > 217800 * 128 and out["parallel"] == out["plain"]
