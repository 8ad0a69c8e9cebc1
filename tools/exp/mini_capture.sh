R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cap; mkdir -p $O; cd $R
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_short.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MfmaUtil $O/b256
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 500 --warmup 100 --no-cpu-baseline --profile-run --replay-size 100000 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 60 --warmup 70 --no-cpu-baseline --profile-run --replay-size 100000 > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b256 -- python $R/bench.py --batch-size 256 --num-actions 3 --steps 200 --warmup 60 --no-cpu-baseline --profile-run --replay-size 100000 > $O/b256.log 2>&1
tail -c 300 $O/bench_short.json
