"""Post-process two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/*_pmc_traffic.json.

Collect on the GPU box (separate passes, --kernel-trace only — never combined with sys/hip traces):
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$c -- \\
      python $REPO/bench.py --steps 60 --warmup 70 --no-cpu-baseline --replay-size 100000
  done
then here:  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/rNN

Counter unit = KB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide
coalesced reads (calibrated on replay_gather_u8: 2 x FETCH = unique frame bytes), so
traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch, mean over the last 2/3 of a kernel's launches.
These are L2<->fabric requests (Infinity-Cache hits included): data fetched by each of the 8 per-XCD L2s counts
once per XCD.
"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_work  # noqa: E402

from bench import ROCPROF_MATCH as MATCH  # noqa: E402

NAMES = ["conv1_fwd(gather+norm+conv+relu)", "conv2_fwd", "conv3_fwd", "fc4_fwd(splitK)", "head(fc5+td+delta)",
         "fc4_dgrad", "fc4_wgrad", "conv3_dgrad", "conv3_wgrad", "conv2_dgrad", "conv2_wgrad",
         "conv1_wgrad", "update(reduce+fc5wgrad+rmsprop)", "rccl_allreduce", "replay_gather_u8", "prep(idx+meta)",
         "bwd3(conv3_dgrad+conv3_wgrad+fc4_wgrad)", "bwd2(conv2_dgrad+conv2_wgrad+fc4_wgrad)", "bwd1(conv1_wgrad+fc4_wgrad)"]   # kernels.h order


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert f, "no counter_collection.csv under " + d
    f.sort(key=os.path.getmtime, reverse=True)                  # the most recent capture in that directory
    # one kernel name can run at several grid sizes in a bench run (the standalone gather: B = 32 and the B = 4096
    # replay_gather_large row): keep the launches of the most frequent grid = the per-step shape
    by_grid = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == counter:
            by_grid[row["Kernel_Name"]][row.get("Grid_Size", "")].append(float(row["Counter_Value"]))
    vals = defaultdict(list)
    for k, g in by_grid.items():
        vals[k] = max(g.values(), key=len)
    return vals


def main():
    fdir, wdir, prefix = sys.argv[1:4]
    B, A = 32, 4
    work = kernel_work(B, A)
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    for name, vals, c in (("FETCH_SIZE", fetch, "FETCH"), ("WRITE_SIZE", write, "WRITE")):
        with open("%s_pmc_%s_per_kernel.csv" % (prefix, name), "w") as o:
            o.write("Kernel_Name,launches,mean_%s_KB,min,max\n" % name)
            for k in sorted(vals):
                v = vals[k]; t = v[len(v) // 3:]
                o.write('"%s",%d,%.2f,%s,%s\n' % (k, len(v), sum(t) / len(t), min(v), max(v)))
    out = {"_method": __doc__.split("Counter unit")[1].strip().replace("\n", " "), "batch_size": B, "num_actions": A, "kernels": {}}
    out["_method"] = "Counter unit " + out["_method"]
    for sub, kid in MATCH:
        fk = [k for k in fetch if sub in k]; wk = [k for k in write if sub in k]
        if not fk or not wk:
            continue
        fv = [x for k in fk for x in fetch[k]]; wv = [x for k in wk for x in write[k]]
        fm = sum(fv[len(fv) // 3:]) / len(fv[len(fv) // 3:]); wm = sum(wv[len(wv) // 3:]) / len(wv[len(wv) // 3:])
        name, by = NAMES[kid], work[kid]["bytes"]
        out["kernels"][name] = {"FETCH_SIZE_KB": round(fm, 1), "WRITE_SIZE_KB": round(wm, 1),
                                "traffic_bytes": int((2 * fm + wm) * 1024), "algorithmic_bytes": int(by)}
    json.dump(out, open(prefix + "_pmc_traffic.json", "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-45s traffic %9d  algorithmic %9d  x%.2f" % (k, v["traffic_bytes"], v["algorithmic_bytes"], v["traffic_bytes"] / max(v["algorithmic_bytes"], 1)))


if __name__ == "__main__":
    main()
