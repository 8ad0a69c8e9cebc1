cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
# the head launch as a prefetcher of fc4_dgrad's weight panel (option pf_w4), B = 32: per-launch times + gradients, then the step rate alternating
( B=32 A=4 timeout 300 python tools/exp/opt_check.py "pf_w4=1" "pf_w4=1" 2>&1 | tail -3 | cut -c1-260
  B=32 A=4 STEPS=3000 REPS=3 timeout 300 python tools/exp/bt_rate.py "" "pf_w4=1" "" "pf_w4=1" "" "pf_w4=1" 2>&1 | tail -6 ) | tee gpurun_out/r5/pf.txt
