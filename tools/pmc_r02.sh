#!/bin/bash
# Round-2 counter passes for the two packet kernels of the bench frame (kernels running alone: one lane, one frame in flight,
# the form bench.py's roofline passes time).  One rocprofv3 run per counter set, never combined with tracing.
# usage: tools/pmc_r02.sh <outdir> [bench args...]     -> <outdir>/summary.txt
set -u
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
export RT_LANES=1
run() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python3 bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-roofline "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE &&
run sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
python3 tools/pmc_summary.py "$OUT" "$OUT/traffic.json" > "$OUT/summary.txt"
python3 tools/pmc_summary.py --counters-json "$OUT" "$OUT/pmc_counters.json"
echo done
