#!/bin/bash
# SQ instruction counters of one configuration (CONFIG=C3 ...), one lane, summed per kernel over the run: tools/pmc_sq_config.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_sq}
mkdir -p "$OUT"
export TMPDIR=/tmp RT_LANES=1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc" -o pmc -- python3 tools/config_times.py > "$OUT/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
out = sys.argv[1]
t = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for path in glob.glob(f"{out}/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        t[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
print(open(f"{out}/pmc.log").read().strip().splitlines()[-1])
for k, c in sorted(t.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    v = c.get("SQ_INSTS_VALU", 0)
    if v < 1e5: continue
    lanes = c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, c.get("SQ_ACTIVE_INST_VALU", 1))
    print(f"{k:44s} calls {calls[k]:4d} VALU {v / 1e6:9.1f} M  SALU {c.get('SQ_INSTS_SALU', 0) / 1e6:8.1f} M  VMEM {c.get('SQ_INSTS_VMEM', 0) / 1e6:7.1f} M  LDS {c.get('SQ_INSTS_LDS', 0) / 1e6:7.1f} M  waves {c.get('SQ_WAVES', 0):9.0f}  thread-cycles/active-inst {lanes:6.2f}")
PY
