#!/bin/bash
# kernels alone (one lane) and the pipelined frame against the number of workgroups per CU of the two level-0 packet kernels
# usage: tools/occ_sweep.sh "near shadow2" ...     e.g. tools/occ_sweep.sh "2 2" "3 2" "5 2"
for cfg in "$@"; do set -- $cfg
  RT_BLOCKS_PER_CU_NEAR=$1 RT_BLOCKS_PER_CU_SHADOW2=$2 RT_LANES=1 timeout -k 10 120 python bench.py --no-cpu-baseline --inflight 1 --steps 30 --min-seconds 0.1 > /tmp/o1.json 2>/tmp/o1.err || { echo "failed $cfg"; tail -3 /tmp/o1.err; exit 1; }
  RT_BLOCKS_PER_CU_NEAR=$1 RT_BLOCKS_PER_CU_SHADOW2=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline --min-seconds 0.3 > /tmp/o2.json 2>/tmp/o2.err || { echo "failed $cfg"; tail -3 /tmp/o2.err; exit 1; }
  python3 -c "
import json
a, b = json.load(open('/tmp/o1.json')), json.load(open('/tmp/o2.json')); r = a['roofline']
print('near/shadow2 WG per CU $1/$2: alone nearest', r['nearest_ms'], 'shadow', r['shadow_ms'], 'one-lane frame', a['ms_per_step'], '| pipelined', b['value'], 'Mrays/s', b['ms_per_step'], 'ms')"
done
