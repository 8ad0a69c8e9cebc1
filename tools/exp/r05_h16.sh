cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_bt.py tests/test_gpu_dqn.py -q -x -m gpu -p no:cacheprovider -k "fp16 or half or h16 or float16 or conv1" > gpurun_out/r5/pytest_h16.log 2>&1
grep -E "passed|failed" gpurun_out/r5/pytest_h16.log | tail -2; grep -n "^E " gpurun_out/r5/pytest_h16.log | head -5
( DATATYPE=float16 B=256 A=3 timeout 300 python tools/exp/opt_check.py "bt:0=1" 2>&1 | tail -2 | cut -c1-300
  DATATYPE=float16 B=256 A=3 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" "bt:0=1" "" "bt:0=1" 2>&1 | tail -4
  B=256 A=3 timeout 300 python tools/exp/opt_check.py "bt:0=2" 2>&1 | tail -2 | cut -c1-100 ) | tee gpurun_out/r5/h16_rate.txt
