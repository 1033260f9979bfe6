#!/bin/bash
# per-kernel averages of the bench frame, kernels alone (one lane, one frame in flight): tools/kstats.sh [lib.so ...]
set -u
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp RT_LANES=1
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(raytracer_amd/lib/libraytracer_amd.so)
for lib in "${libs[@]}"; do
  export RT_LIB_PATH=$PWD/$lib
  v=$(basename $lib .so)
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ks_$v -o p --output-format csv -- python3 bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline --no-roofline --no-baseline-configs > /dev/null 2>&1 || exit 1
  python3 - "$v" <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = glob.glob(f"gpurun_out/ks_{v}/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(v, r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
done
