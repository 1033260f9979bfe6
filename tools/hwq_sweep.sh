# frame rate against the number of hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES, 4 by default) and render lanes
mkdir -p gpurun_out/hwq
out=gpurun_out/hwq/times.txt
for q in 4 8 16; do
  for l in 4 6 8; do
    GPU_MAX_HW_QUEUES=$q RT_LANES=$l timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 60 --warmup 10 > /tmp/b.json || exit 1
    python3 -c "import json;d=json.load(open('/tmp/b.json'));print('queues',$q,'lanes',$l,d['value'],d['ms_per_step'])" >> $out
  done
done
for q in 4 8 16; do
  echo "== queues $q" >> $out
  GPU_MAX_HW_QUEUES=$q CONFIG=C3 timeout -k 10 120 python3 tools/config_times.py >> $out 2>&1 || exit 1
  GPU_MAX_HW_QUEUES=$q CONFIG=C1 C1_SCENES=aeh timeout -k 10 120 python3 tools/config_times.py >> $out 2>&1 || exit 1
done
