"""Tuning + bring-up helper (not product): the block-tile engine (gemm_engine_bt.h) of the throughput regime.
(1) per-launch dispatch-timestamp times of the latency engine (option bt = 0) and of the block-tile engine's built-in shapes;
(2) every menu entry of sdqn_kernels_bt.hip, the K-slab counts of the weight gradients (tps:<l>) and of fc4 forward (s4);
(3) gradients / Q of every variant against the built-in shapes (bit-identical for block shapes, round-off for slab counts).
   python tools/sweep_bt.py [B] [A] [quick]          (GPU box)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = int(sys.argv[2]) if len(sys.argv) > 2 else 3
QUICK = len(sys.argv) > 3
STEPS = 30
NAMES = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 4: "head", 5: "fc4_dgrad", 12: "update", 16: "bwd3", 17: "bwd2", 18: "bwd1"}
ws, wt = xavier_weights(A, 1), xavier_weights(A, 2)
mb = random_minibatch(B, A, 3, reward_range=(-2, 3))
args = make_args(batch_size=B)


def run(opts, check=True):
    net = sd.DeepQNetwork(A, args)
    net.set_weights(wt, 1); net.set_weights(ws, 0)
    for k, v in opts:
        net.set_option(k, v)
    g = q = None
    if check:
        net.set_option("keep_gradients", 1)
        net.train(mb)
        g = [net.get_layer(i, 3) for i in range(5)]
        q = net.last_q()[0]
        net.set_option("keep_gradients", 0)
    for _ in range(5):
        net.train(mb)
    net.profile(True, -1); net.profile_reset()
    for _ in range(STEPS):
        net.train(mb)
    us = {}
    for p in net.profile_read():
        if p["launches"]:
            us[p["id"]] = p["total_ms"] / p["launches"] * 1e3
    net.profile(False)
    return g, q, us


def show(tag, us):
    tot = sum(v for k, v in us.items() if k in NAMES)
    print("%-34s total %6.1f us | " % (tag, tot) + "  ".join("%s %.1f" % (NAMES[k], us[k]) for k in sorted(us) if k in NAMES), flush=True)
    return tot


_, _, us_old = run([("bt", 0), ("s4", 1)], check=False)
show("latency engine (round 3, s4=1)", us_old)
g0, q0, us0 = run([])
base = show("block-tile engine, built-in", us0)

results = {}
def trial(tag, opts, ids):
    try:
        g, q, us = run(opts)
    except Exception as e:
        print("%-34s ERROR %s" % (tag, repr(e)[:160]), flush=True); return
    gerr = max(float(np.abs(a - b).max() / max(1e-6, np.abs(b).max())) for a, b in zip(g, g0))
    qerr = float(np.abs(q - q0).max())
    tot = sum(v for k, v in us.items() if k in NAMES)
    print("%-34s total %6.1f us | %s | grad %.1e q %.1e %s" % (tag, tot, "  ".join("%s %.1f (%.1f)" % (NAMES[k], us.get(k, float("nan")), us0.get(k, float("nan"))) for k in ids),
                                                                gerr, qerr, "ok" if gerr < 2e-5 and qerr < 2e-5 else "MISMATCH"), flush=True)
    results[tag] = dict(total=tot, **{NAMES[k]: us.get(k) for k in ids})

for kid in (1, 2, 3, 5):
    for m in ((6, 7) if QUICK else range(1, 8)):
        trial("%s menu %d" % (NAMES[kid], m), [("bt:%d" % kid, m)], [kid])
for kid in (16, 17):
    for m in ((5, 6) if QUICK else range(1, 7)):
        trial("%s menu %d" % (NAMES[kid], m), [("bt:%d" % kid, m)], [kid])
if QUICK:
    trial("bwd2 menu 5, tps:2 = 8", [("bt:17", 5), ("tps:2", 8)], [17, 12])
    trial("bwd2 menu 5, tps:2 = 10", [("bt:17", 5), ("tps:2", 10)], [17, 12])
    trial("all CPI=2", [("bt:1", 6), ("bt:2", 6), ("bt:16", 5), ("bt:17", 5), ("tps:2", 8)], [1, 2, 16, 17])
    trial("all CPI=2, D=1", [("bt:1", 7), ("bt:2", 7), ("bt:16", 6), ("bt:17", 6), ("tps:2", 8)], [1, 2, 16, 17])
if not QUICK:
    for t3 in (7, 10, 14, 28, 49):
        trial("tps:3 = %d" % t3, [("tps:3", t3)], [16, 12])
    for t2 in (9, 12, 27, 36, 54):
        trial("tps:2 = %d" % t2, [("tps:2", t2)], [17, 12])
    for s4 in (1, 2, 4):
        trial("s4 = %d" % s4, [("s4", s4)], [3, 4])
    for s4, m in ((1, 3), (2, 3), (4, 3), (4, 2), (2, 4)):
        trial("s4 = %d fc4_fwd menu %d" % (s4, m), [("s4", s4), ("bt:3", m)], [3, 4])
    for kid in (1, 2, 3, 5, 16, 17):
        trial("%s on the latency engine" % NAMES[kid], [("bt:%d" % kid, -1)], [kid])
print("RESULTS", json.dumps(results))
