set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final2
cd $R
( time timeout 900 python bench.py > gpurun_out/final2/bench_default.json 2> gpurun_out/final2/bench_default.err ) 2> gpurun_out/final2/bench_default.time
timeout 300 python bench.py --datatype float16 --no-cpu-baseline --steps 3000 --warmup 300 > gpurun_out/final2/bench_fp16.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final2/stats -- python $R/bench.py --steps 500 --warmup 100 --no-cpu-baseline --replay-size 100000 > $R/gpurun_out/final2/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/final2/pmc_$c -- python $R/bench.py --steps 60 --warmup 70 --no-cpu-baseline --replay-size 100000 > $R/gpurun_out/final2/pmc_$c.log 2>&1
done
ls -R $R/gpurun_out/final2 | head -40
du -sh $R/gpurun_out/final2
