cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
( timeout 600 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_bt.py tests/test_gpu_parity_r2.py tests/test_gpu_replay.py -q -x -m "gpu and not experiments" -p no:cacheprovider 2>&1 | tail -3
  B=256 A=3 timeout 300 python tools/exp/opt_check.py "bt:0=1" 2>&1 | tail -2 | cut -c1-80
  B=256 A=3 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" "bt:0=1" "" "bt:0=1" 2>&1 | tail -4 ) | tee gpurun_out/r5/conv1_ab.txt
