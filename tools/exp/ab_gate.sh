# A/B on one box: the product library (gate prefetch in the latency engine) against tools/exp/ab/libsdqn_hip_nogate.so (-DSDQN_GATE_PREFETCH=0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
for rep in 1 2; do
  for cfg in "32 float32 6000" "32 float16 6000" "256 float32 600"; do
    set -- $cfg
    B=$1 DATATYPE=$2 N=$3 timeout 120 python tools/exp/rate_lib.py 2>&1 | tail -1
    LIB=tools/exp/ab/libsdqn_hip_nogate.so B=$1 DATATYPE=$2 N=$3 timeout 120 python tools/exp/rate_lib.py 2>&1 | tail -1
  done
done | tee $O/ab_gate.txt
( time timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu.time
tail -3 $O/pytest_gpu.log | cut -c1-300
