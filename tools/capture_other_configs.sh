R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final4
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final4/b256 -- python $R/bench.py --batch-size 256 --num-actions 3 --steps 200 --warmup 60 --no-cpu-baseline --replay-size 100000 > $R/gpurun_out/final4/b256.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final4/fp16 -- python $R/bench.py --datatype float16 --steps 500 --warmup 100 --no-cpu-baseline --replay-size 100000 > $R/gpurun_out/final4/fp16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final4/bn -- python $R/bench.py --batch-norm --steps 300 --warmup 100 --no-cpu-baseline --replay-size 100000 > $R/gpurun_out/final4/bn.log 2>&1
ls $R/gpurun_out/final4/*/runc/ | head -20
