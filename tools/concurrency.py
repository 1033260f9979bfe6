#!/usr/bin/env python3
"""How much of the bench's timed region has 0, 1, 2, ... kernels in flight: tools/concurrency.py <kernel_trace.csv>
(rocprofv3 --kernel-trace --output-format csv of the default, multi-lane bench.py run)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = []
names = collections.Counter()
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
# the steady part: the middle half of the frames (k_generate launches mark frame shards)
gen = [k for k in ks if "k_generate" in k[2]]
t0, t1 = gen[len(gen) // 4][0], gen[3 * len(gen) // 4][0]
busy = collections.Counter()
by_kernel = collections.Counter()
pts = []
for s, e, n in ks:
    s, e = max(s, t0), min(e, t1)
    if e <= s:
        continue
    pts.append((s, 1)); pts.append((e, -1))
    by_kernel[n.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:40]] += e - s
pts.sort()
depth, last = 0, t0
for t, d in pts:
    busy[depth] += t - last
    last = t
    depth += d
busy[depth] += t1 - last
tot = t1 - t0
frames = len([g for g in gen if t0 <= g[0] < t1])
print(f"window {tot / 1e6:.3f} ms, {frames} k_generate launches ({tot / 1e3 / max(1, frames):.1f} us each)")
for d in sorted(busy):
    print(f"  {d} kernels in flight: {100.0 * busy[d] / tot:5.1f} %")
print("kernel time / wall time:")
for n, v in by_kernel.most_common(8):
    print(f"  {n:42s} {v / tot:6.3f}")
