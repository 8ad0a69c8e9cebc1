# after the last block-tile changes of the round (unconditional ring loads in the weight gradients, fc4_dgrad on the block-tile routine): the
# whole -m gpu suite, the default line, the B = 256 lines and their kernel statistics
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cap; rm -rf $O; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu.time
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_short.json 2>/dev/null
timeout 200 python bench.py --batch-size 256 --num-actions 3 --no-cpu-baseline --steps 600 --warmup 100 --replay-size 200000 > $O/bench_b256.json 2>/dev/null
timeout 200 python bench.py --batch-size 256 --num-actions 6 --no-cpu-baseline --steps 600 --warmup 100 --replay-size 200000 > $O/bench_b256_a6.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b256 -- python $R/bench.py --batch-size 256 --num-actions 3 --steps 200 --warmup 60 --no-cpu-baseline --profile-run --replay-size 100000 > $O/b256.log 2>&1
cd $R
for f in bench_default bench_short bench_b256 bench_b256_a6; do python - $O/$f.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], {k:(v.get("value") if isinstance(v,dict) else None) for k,v in d.items() if k.startswith("config_")}, d.get("kernels_us",{}).get("head(fc5+td+delta)"))
except Exception as e: print(sys.argv[1], "ERR", repr(e)[:200])
P
done
