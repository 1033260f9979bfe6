#!/bin/bash
# A/B of library builds on the bench workload: tools/ab_libs.sh lib1.so lib2.so ...   (prints Mrays/s, frame ms, nearest ms, shadow ms)
set -u
for lib in "$@"; do
  RT_LIB_PATH=$PWD/$lib RT_DEBUG_GRID=1 timeout -k 10 120 python bench.py --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; exit 1; }
  python3 -c "import json,sys;d=json.load(open('/tmp/ab.json'));r=d['roofline'];print(sys.argv[1],d['value'],d['ms_per_step'],r['nearest_ms'],r['shadow_ms'])" "$lib"
  grep "grid.*false, false" /tmp/ab.err | sed 's/^/    /'
done
