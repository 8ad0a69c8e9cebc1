# historical (round 4): uses the experiments build, which left the library in round 5 (tools/exp/experiments_r04.patch on commit 4b59432)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
( timeout 600 python -m pytest tests/test_gpu_bt.py tests/test_gpu_dqn.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
SDQN_LIB_VARIANT=experiments timeout 300 python -m pytest tests/test_gpu_bt.py -m "gpu and experiments" -q -x -p no:cacheprovider -k "ping_pong" 2>&1 | tail -3
STEPS=600 timeout 250 python tools/exp/bt_rate.py "" "bt_xcd=0" "" "bt_xcd=0" 2>&1 | tail -4 ) | tee gpurun_out/ab/final1.txt
