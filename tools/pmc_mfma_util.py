"""Post-process `rocprofv3 --pmc MfmaUtil --kernel-trace` of bench.py (tools/capture_mfma_util.sh) into
profiles/rNN_pmc_mfma_util.json — the north-star's "MFMA utilisation on the FC layers against gfx950 peak".
  python tools/pmc_mfma_util.py gpurun_out/final5/pmc_mfma profiles/r01"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import ROCPROF_MATCH, kernel_work, F32_PEAK  # noqa: E402
from pmc_traffic import NAMES  # noqa: E402


def main():
    d, prefix = sys.argv[1:3]
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
    by_grid = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "MfmaUtil":
            by_grid[r["Kernel_Name"]][r.get("Grid_Size", "")].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    vals, dur = defaultdict(list), defaultdict(list)
    for k, g in by_grid.items():                      # the most frequent grid of a kernel name = its per-step shape
        best = max(g.values(), key=len)
        vals[k] = [x[0] for x in best]; dur[k] = [x[1] for x in best]
    out = {"_method": "rocprofv3 --pmc MfmaUtil --kernel-trace (one pass), bench.py --steps 60 --warmup 70, B=32 A=4 fp32; MfmaUtil = "
                      "sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100 (rocprofv3's derived counter), mean over the last 2/3 of a "
                      "kernel's launches; flop_util = algorithmic FLOP / (kernel duration under the counter pass * 157.3 TFLOP/s)", "kernels": {}}
    w = kernel_work(32, 4)
    for sub, kid in ROCPROF_MATCH:
        ks = [k for k in vals if sub in k]
        if not ks:
            continue
        v = [x for k in ks for x in vals[k]]; t = [x for k in ks for x in dur[k]]
        v, t = v[len(v) // 3:], t[len(t) // 3:]
        us = sum(t) / len(t) / 1e3
        out["kernels"][NAMES[kid]] = {"MfmaUtil_percent": round(sum(v) / len(v), 2), "us_under_counters": round(us, 2),
                                      "flop_util_percent": round(w[kid]["flops"] / (us * 1e-6) / F32_PEAK * 100, 2)}
        print("%-45s MfmaUtil %5.2f %%  %6.2f us  flop-based %5.2f %%" % ((NAMES[kid],) + tuple(out["kernels"][NAMES[kid]].values())))
    json.dump(out, open(prefix + "_pmc_mfma_util.json", "w"), indent=1)


if __name__ == "__main__":
    main()
