#!/bin/bash
# One rocprofv3 counter pass over the bench workload.  usage: tools/pmc_one.sh <outdir> <passname> COUNTER...
set -u
OUT=$1; NAME=$2; shift 2
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$NAME" -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/$NAME.log" 2>&1 || { echo "pass $NAME failed"; tail -5 "$OUT/$NAME.log"; exit 1; }
