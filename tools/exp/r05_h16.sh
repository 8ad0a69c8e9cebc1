cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
( for i in 1 2; do
  SDQN_LIB_PATH=$PWD/tools/exp/_ab/old.so DATATYPE=float16 B=256 A=3 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" 2>&1 | tail -1 | sed 's/^/old /'
  DATATYPE=float16 B=256 A=3 STEPS=400 REPS=3 timeout 300 python tools/exp/bt_rate.py "" 2>&1 | tail -1 | sed 's/^/new /'
done
SDQN_LIB_PATH=$PWD/tools/exp/_ab/old.so DATATYPE=float16 B=256 A=3 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-300 | sed 's/^/old /'
DATATYPE=float16 B=256 A=3 timeout 300 python tools/exp/opt_check.py 2>&1 | tail -1 | cut -c1-300 | sed 's/^/new /' ) | tee gpurun_out/r5/h16_rate.txt
