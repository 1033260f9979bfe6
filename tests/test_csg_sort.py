"""The order in which the device leaves an Intersection's interval list (csg.cpp:199 sorts it with std::sort, which is not stable):
raytracer_amd/csrc/csg_sort.h restates libstdc++'s std::sort -- the function the compiled reference calls -- and this test compares the
two on 260 000 lists of 0 .. 64 elements full of equal keys, including lists built by McIlroy's adversary that drive the introsort into
its heapsort branch.  Host code only (g++); the same header is compiled into the kernels."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_csg_sort_restatement_equals_libstdcxx_sort(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "csg_sort_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "raytracer_amd", "csrc"),
                    os.path.join(ROOT, "tests", "csg_sort_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split()[1:])
    assert int(fields["with_ties_17plus"]) > 100000 and int(fields["heapsorts"]) > 0, r.stdout
