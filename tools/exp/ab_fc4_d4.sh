cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
( B=256 A=3 timeout 300 python tools/exp/opt_check.py "bt:3=8" "bt:5=8" "bt:5=9" "bt:3=1" "bt:5=1" "bt:5=5" "bt:3=8,s4=4" 2>&1 | tail -12 ) | tee gpurun_out/ab/fc4_d4.txt
