cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
( timeout 600 python -m pytest tests/test_gpu_bt.py tests/test_gpu_parity_r2.py -m gpu -q -x -p no:cacheprovider -k "block_tile or batch256 or xcd" 2>&1 | grep -E "passed|failed|Error" | tail -3
STEPS=600 timeout 250 python tools/exp/bt_rate.py "" "bt:5=-1,bt:16=7,bt:17=7" "" "bt:5=-1,bt:16=7,bt:17=7" 2>&1 | tail -4 ) | tee gpurun_out/ab/final2.txt
