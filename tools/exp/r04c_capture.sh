# Round-4 final verification + capture (second session): the whole -m gpu suite, the bench lines whose builds changed since tools/final_capture.sh
# ran (XCD-contiguous block maps at B >= 128, fc4_wgrad tiles of bwd3 at B = 32), kernel statistics, PMC traffic / MfmaUtil of the headline step.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cap; rm -rf $O; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu.time
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_short.json 2>/dev/null
timeout 200 python bench.py --datatype float16 --no-cpu-baseline --steps 3000 --warmup 300 > $O/bench_fp16.json 2>/dev/null
timeout 200 python bench.py --batch-size 256 --num-actions 3 --no-cpu-baseline --steps 600 --warmup 100 --replay-size 200000 > $O/bench_b256.json 2>/dev/null
timeout 200 python bench.py --batch-size 256 --num-actions 6 --no-cpu-baseline --steps 600 --warmup 100 --replay-size 200000 > $O/bench_b256_a6.json 2>/dev/null
timeout 200 python bench.py --batch-size 256 --num-actions 3 --datatype float16 --no-cpu-baseline --steps 600 --warmup 100 --replay-size 200000 > $O/bench_b256_fp16.json 2>/dev/null
timeout 200 python bench.py --single-rank-dp --no-cpu-baseline --steps 2000 --warmup 200 --replay-size 200000 > $O/bench_dp1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 500 --warmup 100 --no-cpu-baseline --profile-run --replay-size 100000 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 60 --warmup 70 --no-cpu-baseline --profile-run --replay-size 100000 > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b256 -- python $R/bench.py --batch-size 256 --num-actions 3 --steps 200 --warmup 60 --no-cpu-baseline --profile-run --replay-size 100000 > $O/b256.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp16_b256 -- python $R/bench.py --datatype float16 --batch-size 256 --num-actions 3 --steps 200 --warmup 60 --no-cpu-baseline --profile-run --replay-size 100000 > $O/fp16_b256.log 2>&1
cd $R
# the kernel trace CSVs of the PMC passes are large: keep the counter tables only
find $O -name "*kernel_trace.csv" -size +8M -delete
for f in bench_default bench_short bench_fp16 bench_b256 bench_b256_a6 bench_b256_fp16 bench_dp1; do python - $O/$f.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], {k:(v.get("value") if isinstance(v,dict) else None) for k,v in d.items() if k.startswith("config_")})
except Exception as e: print(sys.argv[1], "ERR", repr(e)[:200])
P
done
cat $O/pytest_gpu.time | head -2; du -sh $O
