# (SUPERSEDED by tools/exp/r05_call1.sh, which is what round 5 ran: the counter passes below without the experiments-build A/B — that build left the
#  tree in round 5, tools/exp/experiments_r04.patch.)
# Round 5, first gpurun call (prepared at the end of round 4, when the GPU minutes had run out): what the block-tile launches at B = 256 wait for.
# a block-chunk takes ~1 550 cycles of a CU for 1 024 of matrix time; measured away so far — dependent MFMA chains, the
# prefetch depth, a second wave per SIMD, exposed LDS round trips, the work balance (stream-K), the staging path (direct-to-LDS = register ring).
# Left: how fast the access pattern itself is served (128 row pieces of 128 B per chunk and workgroup).  Counter passes are separate runs with
# --kernel-trace only (gpurun refuses --pmc together with the trace domains).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
BENCH="python $R/bench.py --batch-size 256 --num-actions 3 --steps 30 --warmup 30 --no-cpu-baseline --profile-run --replay-size 100000"
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -- $BENCH > $O/pmc_$tag.log 2>&1 || echo "counter set failed: $set" >> $O/failed_sets.txt
done
cd $R
# the direct-to-LDS routine and the stream-K launches of the experiments build against bt_tile, same box
SDQN_LIB_VARIANT=experiments PP_SPECS="bt:1=13 bt:2=13 bt:5=13 bt:3=13,s4=7 bt:16=13 bt:17=13 bt:1=9 bt:2=9" timeout 200 bash tools/exp/ab_pp.sh > $O/ab_exp.txt 2>&1
SDQN_LIB_VARIANT=experiments timeout 200 python -m pytest tests/test_gpu_bt.py -m "gpu and experiments" -q -p no:cacheprovider > $O/pytest_exp.log 2>&1
tail -3 $O/pytest_exp.log; grep -E "defaults|bt:" $O/ab_exp.txt | cut -c1-220; cat $O/failed_sets.txt 2>/dev/null; ls $O
