#!/bin/bash
# A/B of library builds on the bench workload (round 3): kernels alone (one lane, one frame at a time) and the pipelined default.
# usage: tools/ab_round3.sh lib1.so lib2.so ...
set -u
for lib in "$@"; do
  RT_LANES=1 RT_LIB_PATH=$PWD/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --inflight 1 --steps 30 > /tmp/ab1.json 2> /tmp/ab1.err || { echo "$lib failed (one lane)"; tail -3 /tmp/ab1.err; exit 1; }
  RT_LIB_PATH=$PWD/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-roofline > /tmp/ab2.json 2> /tmp/ab2.err || { echo "$lib failed (pipelined)"; tail -3 /tmp/ab2.err; exit 1; }
  python3 - "$lib" <<'PY'
import json, sys
a, b = json.load(open('/tmp/ab1.json')), json.load(open('/tmp/ab2.json'))
r = a['roofline']
print(f"{sys.argv[1]}: pipelined {b['value']:.0f} Mrays/s ({b['ms_per_step']:.4f} ms, min {b['ms_per_step_min']:.4f} max {b['ms_per_step_max']:.4f}); one lane {a['ms_per_step']:.4f} ms, "
      f"nearest {r['nearest_ms']:.4f} shadow {r['shadow_ms']:.4f} ms; steps/packet {r['steps_per_packet_nearest']} / {r['steps_per_packet_shadow']}, busy {r['busy_lane_frac']}")
PY
done
