# experiment: SQ counters per kernel at B = 256 (one PMC pass, kernel-trace only)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4pmc
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/r4pmc/sq -- python $R/bench.py --batch-size 256 --num-actions 3 --steps 30 --warmup 30 --no-cpu-baseline --profile-run --replay-size 100000 > $R/gpurun_out/r4pmc/sq.log 2>&1
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
f = sorted(glob.glob(R + "/gpurun_out/r4pmc/sq/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[r["Kernel_Name"][:70]]["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = open(R + "/gpurun_out/r4pmc/sq_summary.txt", "w")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1]["dur"])):
    n = len(c["SQ_WAVE_CYCLES"])
    if n < 20: continue
    m = {x: sum(v[n // 3:]) / max(1, len(v[n // 3:])) for x, v in c.items()}
    wc = m["SQ_WAVE_CYCLES"]
    line = "%-70s n=%3d dur %6.1f us | wait_any %.2f wait_inst %.2f active %.2f valu %.2f of wave-cycles | mfma_busy/busy %.2f | lds_conflict/wave_cyc %.3f" % (
        k, n, m["dur"] / 8 / 1e3, m["SQ_WAIT_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc, m["SQ_ACTIVE_INST_ANY"] / wc, m["SQ_ACTIVE_INST_VALU"] / wc,
        m["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1, m["SQ_BUSY_CYCLES"]) , m["SQ_LDS_BANK_CONFLICT"] / wc)
    print(line); out.write(line + "\n")
PY
