#!/bin/bash
# sweep of the nearest-hit packet kernel's grid cap (workgroups per CU) on the default, multi-lane bench run
for b in ${BLKS:-1 2 3 5}; do
  export RT_BLOCKS_PER_CU_NEAR=$b
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 --warmup 5 > /tmp/b.json || exit 1
  python -c "import json;d=json.load(open('/tmp/b.json'));r=d['roofline'];print('near blocks per CU',$b,d['value'],d['ms_per_step'],d['single_frame_latency_ms'],r['nearest_ms'],r['shadow_ms'])"
done
