for w in 5 5 50 300 2000; do
  timeout 120 python bench.py --steps 20 --warmup $w --no-cpu-baseline --replay-size 100000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps 20 warmup $w:', d['value'], d['ms_per_step'])"
done
for k in 100 1000; do
  timeout 120 python bench.py --steps $k --warmup 5 --no-cpu-baseline --replay-size 100000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $k warmup 5:', d['value'], d['ms_per_step'])"
done
