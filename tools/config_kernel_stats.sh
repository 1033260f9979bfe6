#!/bin/bash
# Per-kernel totals of one frame loop of C3, C4 and C5 (tools/config_times.py), kernels alone (RT_LANES=1): rocprofv3 --kernel-trace --stats.
# usage (GPU box): bash tools/config_kernel_stats.sh > gpurun_out/r02_config_kernel_stats.txt
set -e
export TMPDIR=/tmp RT_LANES=1
for cfg in C3 C4 C5; do
  export CONFIG=$cfg C5_GRID=1581 C5_AA=4
  rm -rf gpurun_out/cks_$cfg
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d gpurun_out/cks_$cfg -o p --output-format csv -- python3 tools/config_times.py > gpurun_out/cks_$cfg.log 2>&1
  echo "== $cfg (one lane; $(grep -v amdgpu.ids gpurun_out/cks_$cfg.log | grep "ms/frame" | tail -1))"
  python3 - $cfg <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/cks_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
print(f"{'kernel':62s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for r in list(csv.DictReader(open(f)))[:10]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{n[:62]:62s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} {float(r['Percentage']):6.2f}")
PY
done
