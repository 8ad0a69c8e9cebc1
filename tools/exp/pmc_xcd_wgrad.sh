# PMC traffic of the step with the XCD-contiguous tile map on the weight-gradient launches (bwd1: conv1_wgrad; bwd2: conv2_dgrad + conv2_wgrad)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xcdw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  SDQN_BENCH_OPTS="xcd:18=3,xcd:17=7" timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 60 --warmup 70 --no-cpu-baseline --profile-run --replay-size 100000 > $O/pmc_$c.log 2>&1
done
du -sh $O
