# rocprofv3 kernel-stats capture of the default step only (tools/final_capture.sh's stats leg), to a numbered directory
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cap_stats$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 500 --warmup 100 --no-cpu-baseline --profile-run --replay-size 100000 > $O/stats.log 2>&1
python - <<PY
import csv, glob
f = sorted(glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.reader(open(f)))[1:4]: print(r[0][:60], r[3])
PY
