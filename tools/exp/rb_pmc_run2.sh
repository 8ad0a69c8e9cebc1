cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2f
for cfg in "" "nw:1=8" "rb:1=2" "rb:1=4"; do
  tag=$(echo "x$cfg" | tr ':=,' '___')
  timeout 80 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r2f/a_$tag -- python $R/tools/rb_probe.py "$cfg" 6 > $R/gpurun_out/r2f/a_$tag.log 2>&1
  timeout 80 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r2f/b_$tag -- python $R/tools/rb_probe.py "$cfg" 6 > $R/gpurun_out/r2f/b_$tag.log 2>&1
done
grep -l "exceeds" $R/gpurun_out/r2f/*.log
ls $R/gpurun_out/r2f
