"""Post-process tools/exp/r06_pmc_mall.sh: per kernel of the step, L2 -> fabric read requests (TCC_EA0_RDREQ), how many of them are
destined for DRAM (TCC_EA0_RDREQ_DRAM), bytes by request size (the FETCH_SIZE expression: 128 B x BUBBLE + 64 B x (RDREQ - BUBBLE - 32B) +
32 B x RDREQ_32B), L2 hit rate and write requests — means over the last 2/3 of a kernel's launches, next to the algorithmic bytes."""
import csv, glob, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bench import kernel_work, ROCPROF_MATCH as MATCH
from tools.pmc_traffic import NAMES


def load(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not f:
        return {}
    by = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
    for row in csv.DictReader(open(f[-1])):
        by[row["Kernel_Name"]][row.get("Grid_Size", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, g in by.items():
        best = max(g.values(), key=lambda c: len(next(iter(c.values()))))
        out[k] = {c: sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) for c, v in best.items()}
    return out


def main():
    o = sys.argv[1]
    for name, B, A in (("b32", 32, 4), ("b256", 256, 3)):
        rd, hit = load(os.path.join(o, name + "_rd")), load(os.path.join(o, name + "_hit"))
        work = kernel_work(B, A)
        print("== B = %d" % B)
        print("%-46s %10s %10s %7s %11s %11s %7s %8s %10s %10s" % ("kernel", "RDREQ", "RDREQ_DRAM", "dram/all", "read bytes", "algo bytes", "x algo", "L2 hit", "WRREQ", "WRREQ_DRAM"))
        match = MATCH if B == 32 else [("conv1_bf16_rows2_kernel", 0), ("conv_ss_kernel<sdqn::ss::Cfg<20, 20, 32, 4, 4, 2, 9, 9, 2,", 1),
                                       ("conv_ss_kernel<sdqn::ss::Cfg<9, 9, 64, 3, 3, 1, 7, 7, 2,", 2), ("Fc4FwdWT", 3), ("head_kernel", 4),
                                       ("bt_kernel<sdqn::BtCfg<sdqn::Fc4DgradWT", 5), ("bt_multi_kernel<sdqn::BtCfg<sdqn::Conv3DgradWT", 16),
                                       ("bt_multi_kernel<sdqn::BtCfg<sdqn::NoProblem", 17), ("c1w_bt2_kernel", 18), ("update_kernel", 12)]
        for sub, kid in match:
            ks = [k for k in rd if sub in k]
            if not ks:
                continue
            r = rd[ks[0]]; h = hit.get(ks[0], {})
            req, dram, r32, bub = (r.get(c, 0.0) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"))
            rbytes = bub * 128 + (req - bub - r32) * 64 + r32 * 32
            hr = h.get("TCC_HIT_sum", 0.0) / max(h.get("TCC_HIT_sum", 0.0) + h.get("TCC_MISS_sum", 0.0), 1.0)
            print("%-46s %10.0f %10.0f %7.3f %11.0f %11.0f %7.2f %8.3f %10.0f %10.0f" % (NAMES[kid], req, dram, dram / max(req, 1), rbytes, work[kid]["bytes"], rbytes / max(work[kid]["bytes"], 1),
                                                                                          hr, h.get("TCC_EA0_WRREQ_sum", 0.0), h.get("TCC_EA0_WRREQ_DRAM_sum", 0.0)))


if __name__ == "__main__":
    main()
