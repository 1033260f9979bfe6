#!/bin/bash
# Round-3 evidence for the CURRENT sources, collected on the GPU box (kernels alone: RT_LANES=1, one frame in flight):
#   stage pmc    counter passes + kernel trace for every BASELINE configuration's frame loop
#                (C2: bench.py's frame; C3 / C4 / C5: tools/config_times.py)          -> gpurun_out/r03/{pmc,trace}_<CFG>
#                then tools/config_roofline.py -> gpurun_out/r03/pmc_counters.json (bench.py's numerators), config_roofline.json
#   (in the container: python3 tools/install_evidence.py -- gpurun_out/ does not travel back to the box, profiles/ does)
#   stage bench  python3 bench.py with those counters (profiles/r03_pmc_counters.json) -> gpurun_out/r03/bench_c2.json; tools/bench_configs.py -> gpurun_out/r03/configs.json
# One rocprofv3 run per counter set, never combined with tracing.  Afterwards, in the container: python3 tools/install_evidence.py
# usage: tools/evidence_r03.sh pmc [C2 C3 C4 C5] | bench
set -u
STAGE=${1:-pmc}; shift || true
OUT=gpurun_out/r03
mkdir -p "$OUT"
export TMPDIR=/tmp
pass() {   # cfg, name, counters...
  local cfg=$1 name=$2; shift 2
  local d="$OUT/pmc_$cfg/$name"
  rm -rf "$d"; mkdir -p "$d"
  if [ "$cfg" = C2 ]; then
    RT_LANES=1 timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$d" -o pmc -- python3 bench.py --steps 3 --warmup 1 --inflight 1 --min-seconds 0 --no-cpu-baseline --no-roofline > "$d.log" 2>&1
  else
    RT_LANES=1 CONFIG=$cfg C5_GRID=1581 C5_AA=4 timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$d" -o pmc -- python3 tools/config_times.py > "$d.log" 2>&1
  fi
  local rc=$?
  if [ $rc -ne 0 ]; then echo "pass $cfg/$name failed (rc $rc)"; tail -5 "$d.log"; return 1; fi
  echo "pass $cfg/$name ok"
}
trace() {
  local cfg=$1 d="$OUT/trace_$1"
  rm -rf "$d"; mkdir -p "$d"
  if [ "$cfg" = C2 ]; then
    RT_LANES=1 timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$d" -o t --output-format csv -- python3 bench.py --steps 10 --warmup 2 --inflight 1 --min-seconds 0 --no-cpu-baseline --no-roofline > "$d.log" 2>&1
  else
    RT_LANES=1 CONFIG=$cfg C5_GRID=1581 C5_AA=4 timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$d" -o t --output-format csv -- python3 tools/config_times.py > "$d.log" 2>&1
  fi
  local rc=$?
  if [ $rc -ne 0 ]; then echo "trace $cfg failed (rc $rc)"; tail -5 "$d.log"; return 1; fi
  echo "trace $cfg ok"
}
if [ "$STAGE" = pmc ]; then
  CFGS=${*:-C2 C3 C4 C5}
  for cfg in $CFGS; do
    pass $cfg sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
    pass $cfg sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE &&
    pass $cfg tcc TCC_HIT_sum TCC_MISS_sum &&
    pass $cfg fetch FETCH_SIZE &&
    pass $cfg write WRITE_SIZE &&
    trace $cfg || exit 1
  done
  python3 tools/config_roofline.py "$OUT" || exit 1
  echo "pmc stage done"
elif [ "$STAGE" = bench ]; then
  timeout -k 10 300 python3 bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err" || { echo "bench failed"; tail -5 "$OUT/bench_c2.err"; exit 1; }
  cat "$OUT/bench_c2.json"
  timeout -k 10 600 python3 tools/bench_configs.py "$OUT/configs.json" > "$OUT/configs.log" 2>&1 || { echo "configs failed"; tail -5 "$OUT/configs.log"; exit 1; }
  grep -v amdgpu.ids "$OUT/configs.log" | tail -20
fi
