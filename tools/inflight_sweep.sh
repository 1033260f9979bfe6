for d in 2 3 4; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 --warmup 5 --inflight $d > /tmp/b.json || exit 1
  python -c "import json;d=json.load(open('/tmp/b.json'));print('inflight',$d,d['value'],d['ms_per_step'])"
done
for l in 3 2; do
  RT_LANES=$l timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 --warmup 5 > /tmp/b.json || exit 1
  python -c "import json;d=json.load(open('/tmp/b.json'));print('lanes',$l,d['value'],d['ms_per_step'])"
done
