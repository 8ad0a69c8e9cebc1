cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
# same-box A/B of two builds at B = 256 (and the B = 32 headline untouched?): libsdqn_hip_prev.so = the build before the change (SDQN_LIB_PATH), alternating
P=$GRAFT_REPO_ROOT/simple_dqn_amd/libsdqn_hip_prev.so
( for i in 1 2 3; do
    echo "--- prev"; SDQN_LIB_PATH=$P B=256 A=4 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-250
    echo "--- new";  B=256 A=4 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-250
  done
  for i in 1 2; do
  echo "--- prev rate"; SDQN_LIB_PATH=$P B=256 A=4 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" 2>&1 | tail -1
  echo "--- new rate";  B=256 A=4 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" 2>&1 | tail -1
  done
  echo "--- B=128 prev/new"; SDQN_LIB_PATH=$P B=128 A=4 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-250; B=128 A=4 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-250
  timeout 600 python -m pytest tests/test_gpu_bt.py tests/test_gpu_parity_r2.py -q -x -m gpu -p no:cacheprovider -k "conv1 or fused or large_batch or b256 or B256" 2>&1 | tail -2
) | tee gpurun_out/r5/ab_prev.txt
