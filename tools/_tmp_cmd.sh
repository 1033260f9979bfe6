set -e
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_suite.log 2>&1 || { tail -40 gpurun_out/gpu_suite.log; exit 1; }
tail -2 gpurun_out/gpu_suite.log
CONFIG=C4 timeout -k 10 120 python3 tools/config_times.py 2>&1 | grep -v amdgpu.ids
CONFIG=C4 timeout -k 10 120 python3 tools/config_times.py 2>&1 | grep -v amdgpu.ids
CONFIG=C1 C1_SCENES=m timeout -k 10 120 python3 tools/config_times.py 2>&1 | grep -v amdgpu.ids
