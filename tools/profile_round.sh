#!/bin/bash
# One round's evidence on the GPU box: kernel trace of the bench command, HBM-traffic counter passes, then the plain bench.
# usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the profiled runs use one lane and one frame in flight: every kernel then runs alone over the whole frame, which is
# what bench.py's roofline passes time with hipEvents (the default run overlaps shards and frames)
export RT_LANES=1
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python3 bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
DB=$(ls "$OUT"/trace/*.db "$OUT"/trace/*/*.db 2>/dev/null | head -1)
python3 tools/rocpd_summary.py "$DB" > "$OUT/kernel_stats.txt" || exit 1
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc/$pass" -o pmc -- python3 bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > "$OUT/pmc_$pass.log" 2>&1 || { echo "pmc $pass failed"; tail -5 "$OUT/pmc_$pass.log"; exit 1; }
done
python3 tools/pmc_summary.py "$OUT/pmc" "$OUT/traffic.json" > "$OUT/pmc_summary.txt" || exit 1
unset RT_LANES
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
