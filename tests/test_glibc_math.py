"""raytracer_amd/csrc/glibc_math.h restates glibc's atan2f and asinf (the device picks bump-map texels with them, bump.cpp:14-23, and
the reference's choice of texel is glibc's): compiled with g++ and compared with the host's libm bit for bit on 20 M random arguments
and every float in [-1, 1].  Host code only; the same header is compiled into the kernels."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_glibc_mathment_equals_glibc(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "glibc_math_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "raytracer_amd", "csrc"),
                    os.path.join(ROOT, "tests", "glibc_math_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "atan2f_mismatches=0 asinf_mismatches=0 asinf_exhaustive_mismatches=0" in r.stdout
