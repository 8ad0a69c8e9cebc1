# rocprofv3 kernel + memory-copy trace of the reference's loop body net.train(mem.getMinibatch()) (tools/exp/tuple_api_rate.py): which packets
# a tuple iteration puts on the stream.  ONLY_TUPLE=1 stops the tool after the ReplayMemory leg (no pageable-tuple leg in the trace).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tuple_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ONLY_TUPLE=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/t -- python $R/tools/exp/tuple_api_rate.py > $O/run.log 2>&1
tail -2 $O/run.log
python - <<PY
import csv, glob
for pat in ("*kernel_stats.csv", "*memory_copy_stats.csv", "*domain_stats.csv"):
    fs = sorted(glob.glob("$O/t/**/" + pat, recursive=True))
    if not fs: print("no", pat); continue
    print("==", fs[-1].split("/")[-1])
    for r in list(csv.reader(open(fs[-1])))[:16]: print(", ".join(x[:70] for x in r[:5]))
PY
