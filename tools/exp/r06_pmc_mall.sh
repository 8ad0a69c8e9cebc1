# Round 6: how much of the L2 <-> fabric READ traffic of the step's kernels is destined for DRAM — one --pmc pass per
# counter group (kernel-trace only, never combined with other trace domains), B = 32 (headline step) and B = 256.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_mall
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "b32 --steps 60 --warmup 70" "b256 --batch-size 256 --num-actions 3 --steps 40 --warmup 30"; do
  set -- $cfg; name=$1; shift
  timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d $O/${name}_rd -- \
    python $R/bench.py "$@" --no-cpu-baseline --replay-size 100000 --no-b256 --no-fp16-leg > $O/${name}_rd.log 2>&1
  timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum --kernel-trace --output-format csv -d $O/${name}_hit -- \
    python $R/bench.py "$@" --no-cpu-baseline --replay-size 100000 --no-b256 --no-fp16-leg > $O/${name}_hit.log 2>&1
done
find $O -name "*counter_collection.csv" | head
python $R/tools/exp/r06_pmc_mall.py $O > $O/r06_pmc_mall.txt 2>&1
tail -60 $O/r06_pmc_mall.txt
