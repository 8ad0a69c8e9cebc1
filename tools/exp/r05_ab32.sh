cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
( B=32 A=4 timeout 300 python tools/exp/opt_check.py "wkt=1" "wkt=0" "wkt=1" 2>&1 | tail -4 | cut -c1-230
  B=32 A=4 STEPS=3000 REPS=3 timeout 300 python tools/exp/bt_rate.py "" "wkt=1" "" "wkt=1" 2>&1 | tail -4 ) | tee gpurun_out/r5/ab32.txt
