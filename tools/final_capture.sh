# Regenerates the round's evidence under gpurun_out/cap/ on the GPU box (then tools/collect_profiles.py copies the
# summaries into profiles/rNN_*).  Every step has its own timeout; PMC passes are separate and --kernel-trace only.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cap
mkdir -p $O
cd $R
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_short.json 2>/dev/null
timeout 200 python bench.py --datatype float16 --no-cpu-baseline --steps 3000 --warmup 300 > $O/bench_fp16.json 2>/dev/null
timeout 200 python bench.py --datatype float16 --num-actions 6 --no-cpu-baseline --steps 3000 --warmup 300 --replay-size 200000 > $O/bench_fp16_a6.json 2>/dev/null
timeout 200 python bench.py --batch-size 256 --num-actions 3 --no-cpu-baseline --steps 600 --warmup 100 --replay-size 200000 > $O/bench_b256.json 2>/dev/null
timeout 200 python bench.py --batch-size 256 --num-actions 6 --no-cpu-baseline --steps 600 --warmup 100 --replay-size 200000 > $O/bench_b256_a6.json 2>/dev/null
timeout 200 python bench.py --batch-size 256 --num-actions 3 --datatype float16 --no-cpu-baseline --steps 600 --warmup 100 --replay-size 200000 > $O/bench_b256_fp16.json 2>/dev/null
timeout 200 python bench.py --single-rank-dp --no-cpu-baseline --steps 2000 --warmup 200 --replay-size 200000 > $O/bench_dp1.json 2>/dev/null
# the whole agent loop on the synthetic environment (README table row) and the reference-style tuple-API loop
timeout 300 python -m simple_dqn_amd.main --replay_size 100000 --random_steps 5000 --train_steps 40000 --test_steps 20000 --epochs 1 --csv_file $O/agent_loop.csv > $O/agent_loop.log 2>&1
timeout 300 python -m simple_dqn_amd.main --synthetic_frame_pool 0 --replay_size 100000 --random_steps 5000 --train_steps 40000 --test_steps 20000 --epochs 1 > $O/agent_loop_pool0.log 2>&1
timeout 200 python tools/exp/act_stamps.py > $O/act_stamps.txt 2>&1
timeout 200 python tools/exp/tuple_api_rate.py > $O/tuple_api_rate.txt 2>&1
timeout 200 python tools/generic_rate.py > $O/generic_rate.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 500 --warmup 100 --no-cpu-baseline --profile-run --replay-size 100000 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 60 --warmup 70 --no-cpu-baseline --profile-run --replay-size 100000 > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b256 -- python $R/bench.py --batch-size 256 --num-actions 3 --steps 200 --warmup 60 --no-cpu-baseline --profile-run --replay-size 100000 > $O/b256.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp16 -- python $R/bench.py --datatype float16 --steps 500 --warmup 100 --no-cpu-baseline --profile-run --replay-size 100000 > $O/fp16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp16_b256 -- python $R/bench.py --datatype float16 --batch-size 256 --num-actions 3 --steps 200 --warmup 60 --no-cpu-baseline --profile-run --replay-size 100000 > $O/fp16_b256.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bn -- python $R/bench.py --batch-norm --steps 300 --warmup 100 --no-cpu-baseline --profile-run --replay-size 100000 > $O/bn.log 2>&1
ls -R $O | head -60
du -sh $O
