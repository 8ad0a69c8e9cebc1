cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
( B=256 A=3 timeout 300 python tools/exp/opt_check.py "bt:0=3" "bt:0=4" "bt:0=0" "bt:0=3" "bt:0=4" 2>&1 | tail -6 | cut -c1-80
  B=256 A=3 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" "bt:0=3" "bt:0=4" "" "bt:0=3" "bt:0=4" 2>&1 | tail -6 ) | tee gpurun_out/r5/conv1_ab4.txt
