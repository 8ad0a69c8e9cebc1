"""[historical, rounds 2-4: the float32 register-blocked routine and its option "rb:<id>" left the library in round 5 — runs on the
tree of tools/exp/experiments_r04.patch]  Tuning + bring-up helper (not product): the register-blocked tile routine (gemm_engine_rb.h) at B = 256.
For every kernel id and menu entry of sdqn_kernels_rb.hip: (1) one train step from identical weights on an identical
minibatch vs the unblocked routine — gradients of all layers and Q must agree to fp32 round-off (the two routines differ
only in how K is split over waves, i.e. in summation grouping); (2) HIP-event time per launch over STEPS steps.
   python tools/sweep_rb.py [B] [A]          (GPU box)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = int(sys.argv[2]) if len(sys.argv) > 2 else 3
DT = os.environ.get("DATATYPE", "float32")
STEPS = 12
NAMES = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 5: "fc4_dgrad", 6: "fc4_wgrad", 7: "conv3_dgrad",
         8: "conv3_wgrad", 9: "conv2_dgrad", 10: "conv2_wgrad", 11: "conv1_wgrad"}
MENU = {0: 4, 1: 5, 2: 5, 3: 4, 5: 4, 6: 4, 7: 4, 8: 4, 9: 4, 10: 4, 11: 4}
if DT == "float16":                      # gemm_tile_hb menus (forward / dgrad only)
    MENU = {0: 3, 1: 5, 2: 5, 3: 4, 5: 4, 7: 4, 9: 3}
TPS = {8: ("tps:3", [25, 13, 7]), 10: ("tps:2", [41, 21, 11]), 11: ("tps:1", [100, 50, 25])}     # chunks per slab to try with the blocked wgrads

ws, wt = xavier_weights(A, 1), xavier_weights(A, 2)
mb = random_minibatch(B, A, 3, reward_range=(-2, 3))
args = make_args(batch_size=B, datatype=DT)


def run(opts, time_it=True):
    net = sd.DeepQNetwork(A, args)
    net.set_weights(wt, 1); net.set_weights(ws, 0)
    net.set_option("keep_gradients", 1)
    net.set_option("fused_launches", 0)
    for k, v in opts:
        net.set_option(k, v)
    net.train(mb)
    g = [net.get_layer(i, 3) for i in range(5)]
    q = net.last_q()[0]
    us = {}
    if time_it:
        net.set_option("keep_gradients", 0)
        for _ in range(3):
            net.train(mb)
        net.profile(True, -1); net.profile_reset()
        for _ in range(STEPS):
            net.train(mb)
        for p in net.profile_read():
            if p["launches"]:
                us[p["id"]] = p["total_ms"] / p["launches"] * 1e3
        net.profile(False)
    return g, q, us


g0, q0, us0 = run([])
print("baseline (unblocked) us per launch:", {NAMES.get(k, k): round(v, 1) for k, v in us0.items()}, flush=True)
best = {}
for kid, n in MENU.items():
    for m in range(1, n + 1):
        tps_opts = [None] + ([(TPS[kid][0], t) for t in TPS[kid][1]] if (kid in TPS and DT == "float32") else [])
        for tp in tps_opts:
            opts = [("rb:%d" % kid, m)] + ([tp] if tp else [])
            try:
                g, q, us = run(opts)
            except Exception as e:
                print(NAMES[kid], m, tp, "ERROR", repr(e)[:200], flush=True); continue
            gerr = max(float(np.abs(a - b).max() / max(1e-6, np.abs(b).max())) for a, b in zip(g, g0))
            qerr = float(np.abs(q - q0).max())
            ok = gerr < (2e-2 if DT == "float16" else 2e-5) and qerr < (3e-3 if DT == "float16" else 2e-5)     # fp16: a different K split moves half roundings
            t = us.get(kid, float("nan"))
            print("%-12s menu %d %-12s  %7.1f us  (unblocked %7.1f)  grad rel err %.1e  q err %.1e  %s"
                  % (NAMES[kid], m, tp or "", t, us0.get(kid, float("nan")), gerr, qerr, "ok" if ok else "MISMATCH"), flush=True)
            if ok and (kid not in best or t < best[kid][0]):
                best[kid] = (t, m, tp)
print("BEST", json.dumps({NAMES[k]: v for k, v in best.items()}))
print("sum unblocked %.1f us, sum best %.1f us" % (sum(us0.get(k, 0) for k in MENU), sum(best[k][0] if k in best else us0.get(k, 0) for k in MENU)))
