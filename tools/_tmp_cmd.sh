set -e
for i in 1 2 3; do
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/q.json
python3 -c "import json;d=json.load(open('gpurun_out/q.json'));print(d['value'],d['ms_per_step'])"
done
