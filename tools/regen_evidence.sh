#!/bin/bash
# Everything profiles/ needs from the GPU for the current sources, in the order bench.py wants it (run on the GPU box):
#   1. counter passes of the bench frame, kernels alone (tools/pmc_r02.sh)      -> gpurun_out/r02_pmc
#   2. kernel trace of the same run                                              -> gpurun_out/r02_trace
# then here: python3 tools/install_evidence.py, and on the box again: python3 bench.py > gpurun_out/r02_bench.json and
# python3 tools/bench_configs.py gpurun_out/r02_configs.json (bench.py takes its counters from profiles/, tied to the source hash).
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/r02_pmc gpurun_out/r02_trace
bash tools/pmc_r02.sh gpurun_out/r02_pmc
mkdir -p gpurun_out/r02_trace
RT_LANES=1 timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r02_trace -o bench --output-format csv -- python3 bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline --no-baseline-configs > gpurun_out/r02_trace.log 2>&1
ls gpurun_out/r02_trace
