#!/bin/bash
# kernels of the last frame of a bundled scene on one lane, with durations and the gaps between them: tools/dbg/frame_timeline.sh <scene letter>
set -e
export TMPDIR=/tmp RT_LANES=1 CONFIG=C1 C1_SCENES=$1
rm -rf gpurun_out/trace_$1
timeout -s KILL 200 rocprofv3 --kernel-trace -d gpurun_out/trace_$1 -o t --output-format csv -- python3 tools/config_times.py > gpurun_out/trace_$1.log 2>&1
python3 - $1 <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/trace_{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"]][-1]
prev_end = None
for r in rows[idx:idx + 60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{name:46s} dur {(e - s) / 1e3:7.2f} us  gap before {gap:6.2f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    prev_end = e
PY
