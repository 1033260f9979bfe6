#!/usr/bin/env python3
"""Timeline of the last frame in a rocprofv3 --kernel-trace csv: per kernel start offset, duration and the gap since the previous kernel ended
(one lane: RT_LANES=1).  usage: tools/timeline.py <dir with *_kernel_trace.csv> [max rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
# last frame: from the last k_generate on
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_generate")]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
busy = 0
out = []
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {name(r)[:60]}")
    busy += e - s
    prev_end = max(prev_end, e)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
print("\n".join(out[:n]))
print(f"... {len(out)} kernels, span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
