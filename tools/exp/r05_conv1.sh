# round 5: conv1 forward at B >= 128: specialised waves (default) vs the 10-wave row-chunk kernel (bt:0=2) vs round 4's staged kernel (bt:0=1), same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
( timeout 300 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_bt.py -q -x -k "conv1 or block_tile" -m gpu -p no:cacheprovider 2>&1 | tail -2
  B=256 A=3 timeout 300 python tools/exp/opt_check.py "bt:0=2" "bt:0=1" 2>&1 | tail -3 | cut -c1-80
  B=256 A=3 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" "bt:0=2" "bt:0=1" "" "bt:0=2" "bt:0=1" 2>&1 | tail -6 ) | tee gpurun_out/r5/conv1_ab3.txt
