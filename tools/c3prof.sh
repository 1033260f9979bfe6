#!/bin/bash
# per-kernel times of one configuration (CONFIG, default C3) on one lane: tools/c3prof.sh [lib.so ...]
set -u
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp RT_LANES=1 CONFIG=${CONFIG:-C3}
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(raytracer_amd/lib/libraytracer_amd.so)
for lib in "${libs[@]}"; do
  export RT_LIB_PATH=$PWD/$lib
  v=$(basename $lib .so)
  timeout -k 10 200 python3 tools/config_times.py || exit 1
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$v -o p --output-format csv -- python3 tools/config_times.py > /dev/null 2>&1 || exit 1
  python3 - "$v" <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_{v}/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(v, r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
done
