#!/bin/bash
# per-kernel times of the C5 frame, kernels alone (one lane): tools/c5_kernels.sh [lib.so]   (env passes through, e.g. RT_NEAREST2_MIN_SPP=0)
export TMPDIR=/tmp RT_LANES=1 CONFIG=${CONFIG:-C5} C5_GRID=${C5_GRID:-1581} C5_AA=${C5_AA:-4}
[ -n "${1:-}" ] && export RT_LIB_PATH=$PWD/$1
d=gpurun_out/c5k; rm -rf $d
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $d -o p --output-format csv -- python3 tools/config_times.py > $d.log 2>&1 || { echo trace failed; tail -3 $d.log; exit 1; }
grep "ms/frame" $d.log | tail -1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c5k/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"  {n[:60]:60s} {r['Calls']:>5s} {float(r['TotalDurationNs']) / 1e6:10.3f} ms {float(r['AverageNs']) / 1e3:10.2f} us {float(r['Percentage']):6.2f}%")
PY
