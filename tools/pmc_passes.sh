#!/bin/bash
# Collects rocprofv3 counters for the bench workload in separate passes (one run per counter set; never with tracing).
# usage: tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_VMEM &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
echo done
