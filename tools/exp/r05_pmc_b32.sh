# round 5: TCP / TCC / TA counters on the B = 32 step (what a latency-engine tile's load ISSUE waits for: tools/landing_hist.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 60 --warmup 70 --no-cpu-baseline --profile-run --replay-size 100000"
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_INSTS_LDS SQ_INST_CYCLES_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -- $BENCH > $O/pmc_$tag.log 2>&1 || echo "counter set failed: $set" >> $O/failed_sets.txt
done
cd $R
python tools/exp/r05_pmc_table.py $O > $O/pmc_table.txt 2>&1
cat $O/failed_sets.txt 2>/dev/null; wc -l $O/pmc_table.txt
