cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
( B=256 A=3 STEPS=400 REPS=3 timeout 400 python tools/exp/bt_rate.py "" "tps:1=15" "tps:1=20" "" "tps:1=15" "tps:1=20" "tps:1=5" 2>&1 | tail -7
  B=256 A=3 timeout 300 python tools/exp/opt_check.py "tps:1=15" "tps:1=20" 2>&1 | tail -3 | cut -c1-250 ) | tee gpurun_out/r5/tps.txt
