#!/usr/bin/env python3
"""Wave lifetimes of the packet kernels from an RT_PROF dump (RT_TRACE_FILE): launch ramp, tail, longest packets."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint8)
off = 2 * 4 * 2048 * 8
log = raw[off:off + 2 * 8192 * 32].view(np.uint32).reshape(2, 8192, 8)
for k, name in enumerate(("nearest", "shadow")):
    L = log[k]
    L = L[L[:, 3] > 0].astype(np.int64)
    if not len(L):
        continue
    t0 = L[:, 0].min()
    start, end = (L[:, 0] - t0) / 100.0, (L[:, 1] - t0) / 100.0   # us
    print(f"{name}: {len(L)} waves, kernel span {end.max():.1f} us; start p50/p99/max {np.percentile(start, 50):.1f}/{np.percentile(start, 99):.1f}/{start.max():.1f} us; "
          f"end p1/p10/p50/p90/max {np.percentile(end, 1):.1f}/{np.percentile(end, 10):.1f}/{np.percentile(end, 50):.1f}/{np.percentile(end, 90):.1f}/{end.max():.1f} us")
    print(f"   mean wave lifetime {np.mean(end - start):.1f} us = {100 * np.mean(end - start) / end.max():.0f} % of the span; packets/wave mean {L[:, 3].mean():.1f} max {L[:, 3].max()}; "
          f"longest packet p50/p99/max {np.percentile(L[:, 2], 50) / 100:.1f}/{np.percentile(L[:, 2], 99) / 100:.1f}/{L[:, 2].max() / 100:.1f} us")
    heavy = L[np.argsort(-L[:, 2])[:12]]
    print("   heaviest packets (us, union steps, max nodes of one ray, mean nodes per ray, rays): " +
          "; ".join(f"{h[2] / 100:.0f}/{h[4]}/{h[5]}/{h[6] / max(h[7], 1):.0f}/{h[7]}" for h in heavy))
    print(f"   over the waves' longest packets: union steps p50 {np.percentile(L[:, 4], 50):.0f}, max-ray nodes p50 {np.percentile(L[:, 5], 50):.0f}, us per union step p50 {np.percentile(L[:, 2] / np.maximum(L[:, 4], 1), 50) / 100:.2f}")
