set -e
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_suite.log 2>&1 || { tail -40 gpurun_out/gpu_suite.log; exit 1; }
tail -2 gpurun_out/gpu_suite.log
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 60 --warmup 10 > gpurun_out/q.json
python3 -c "import json;d=json.load(open('gpurun_out/q.json'));print(d['value'],d['ms_per_step'])"
CONFIG=C5 C5_GRID=1581 C5_AA=4 bash tools/c3prof.sh 2>&1 | grep -v amdgpu.ids | head -5
CONFIG=C5 C5_GRID=1581 C5_AA=4 timeout -k 10 120 python3 tools/config_times.py 2>&1 | grep -v amdgpu.ids
bash tools/kstats.sh
