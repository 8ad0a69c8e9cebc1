cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
( timeout 200 python tools/exp/opt_check.py "c3d9=1" "c3d9=1" 2>&1 | tail -4
OPTS='[[("c3d9",1)],[("c3d9",1)]]' N=6000 timeout 200 python tools/ab_options.py 2>&1 | tail -6 ) | tee gpurun_out/ab/c3d9.txt
