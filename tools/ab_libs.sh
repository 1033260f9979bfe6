#!/bin/bash
# A/B of library builds on the bench workload, one lane / one frame at a time so that kernel times add up:
# tools/ab_libs.sh lib1.so lib2.so ...   (prints Mrays/s, frame ms, nearest ms, shadow ms)
set -u
export RT_LANES=${AB_LANES:-1}
for lib in "$@"; do
  RT_LIB_PATH=$PWD/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --inflight ${AB_INFLIGHT:-1} > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; exit 1; }
  python3 -c "import json,sys;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print(sys.argv[1],d['value'],d['ms_per_step'],r['nearest_ms'],r['shadow_ms'])" "$lib"
done
