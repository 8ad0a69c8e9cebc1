cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_replay.py tests/test_tracked_views.py tests/test_cabi.py tests/test_gpu_dqn.py -q -x -m gpu -p no:cacheprovider > gpurun_out/r5/pytest_tuple.log 2>&1
grep -E "passed|failed" gpurun_out/r5/pytest_tuple.log | tail -3; grep -n "^E " gpurun_out/r5/pytest_tuple.log | head -5
timeout 300 python tools/exp/tuple_api_rate.py 2>&1 | grep steps/s | tee gpurun_out/r5/tuple_rate.txt
