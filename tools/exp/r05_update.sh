cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
# same-box A/B of the optimizer pass at B = 256: libsdqn_hip_prev.so = the build before the change (SDQN_LIB_PATH), alternating
P=$GRAFT_REPO_ROOT/simple_dqn_amd/libsdqn_hip_prev.so
( for i in 1 2; do
    echo "--- prev"; SDQN_LIB_PATH=$P B=256 A=4 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-250
    echo "--- new";  B=256 A=4 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-250
  done
  echo "--- prev rate"; SDQN_LIB_PATH=$P B=256 A=4 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" 2>&1 | tail -1
  echo "--- new rate";  B=256 A=4 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" 2>&1 | tail -1
  echo "--- prev rate"; SDQN_LIB_PATH=$P B=256 A=4 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" 2>&1 | tail -1
  echo "--- new rate";  B=256 A=4 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" 2>&1 | tail -1
  echo "--- B=128 prev/new"; SDQN_LIB_PATH=$P B=128 A=4 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-250; B=128 A=4 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-250
) | tee gpurun_out/r5/update_ab.txt
