cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
( B=256 A=3 timeout 400 python tools/exp/opt_check.py $PP_SPECS 2>&1 | tail -16 ) | tee gpurun_out/ab/pp.txt
