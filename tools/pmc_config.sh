#!/bin/bash
# HBM-side traffic of one of the other configurations (CONFIG=C5 ...), one lane: separate FETCH_SIZE / WRITE_SIZE passes
# and a kernel trace of tools/config_times.py.  usage: CONFIG=C5 C5_GRID=1581 tools/pmc_config.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_cfg}
mkdir -p "$OUT"
export TMPDIR=/tmp RT_LANES=1
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc/$pass" -o pmc -- python3 tools/config_times.py > "$OUT/pmc_$pass.log" 2>&1 || { echo "pmc $pass failed"; tail -5 "$OUT/pmc_$pass.log"; exit 1; }
done
python3 tools/pmc_summary.py "$OUT/pmc" "$OUT/traffic.json" > "$OUT/pmc_summary.txt" || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o cfg --output-format csv -- python3 tools/config_times.py > "$OUT/trace.log" 2>&1 || { echo "trace failed"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
f = glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True)[0]
tr = json.load(open(f"{out}/traffic.json"))
print(open(f"{out}/trace.log").read().strip().splitlines()[-1])
for r in list(csv.DictReader(open(f)))[:6]:
    name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    base = name.split("<")[0]
    t = tr.get(base, {}).get("traffic_bytes_per_launch")
    avg = float(r["AverageNs"])
    print(f"{name:42s} calls {r['Calls']:>4s} avg {avg / 1e3:9.1f} us" + (f"  HBM-side {t / 1e6:8.1f} MB/launch = {t / avg / 1e3:.3f} TB/s" if t else ""))
PY
