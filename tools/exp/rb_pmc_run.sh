cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "" "rb:1=2" "rb:1=4" "rb:3=2"; do
  tag=$(echo "x$cfg" | tr ':=,' '___')
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r2d/pmc_$tag -- python $R/tools/rb_probe.py "$cfg" 6 > $R/gpurun_out/r2d/pmc_$tag.log 2>&1
done
ls -R $R/gpurun_out/r2d | head -30
