# per-kernel instruction-cache counters of the B = 32 step (separate --pmc pass, kernel trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|SQC_INST|SQ_INSTS_VALU |SQ_BUSY_CYCLES|SQ_WAVES " | head -40 > $O/avail.txt
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --steps 60 --warmup 70 --no-cpu-baseline --profile-run --replay-size 100000 > $O/pmc_$n.log 2>&1
done
python - <<'P'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/icache"
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k, d in acc.items():
        print("  %-72s" % k, "  ".join("%s=%.0f (n=%d)" % (c, sum(v[len(v)//3:]) / max(len(v[len(v)//3:]), 1), len(v)) for c, v in d.items()))
P
cat $O/avail.txt | head -20
