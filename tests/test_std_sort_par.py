"""The loader sorts a BVH node's objects with a multi-threaded restatement of libstdc++'s std::sort (raytracer_amd/csrc/host/std_sort_par.h):
the reference calls std::sort (bvh.cpp:188-213), which is not stable, and the order it leaves equal centroids in decides leaf contents and
the DFS order behind every tie.  This test compares the restatement with std::sort itself, element for element, on keys full of ties --
from empty lists to 1.5 M elements, partitions running on threads of their own -- and on inputs built by McIlroy's adversary that take
the introsort into its heapsort branch.  Host code only (g++)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parallel_sort_equals_libstdcxx_sort(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "std_sort_par_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "raytracer_amd", "csrc"),
                    os.path.join(ROOT, "tests", "std_sort_par_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split()[1:])
    assert int(fields["cases"]) > 10000 and int(fields["heapsorts"]) > 0, r.stdout
