"""Identity of the kernel sources a counter file was taken with (bench.py ties profiles/*_pmc_counters.json to the build it describes)."""
import glob
import hashlib
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha():
    """sha256 over the device and host sources of libraytracer_amd.so, in a fixed order."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_ROOT, "raytracer_amd", "csrc", "*.hip")) + glob.glob(os.path.join(_ROOT, "raytracer_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(_ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
