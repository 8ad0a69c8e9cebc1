"""Experiment helper (not product): gradients / Q of one train step under an option set against the defaults, and per-launch times.
   python tools/exp/opt_check.py "c3d9=1" ["opt=v,opt=v" ...]      env: B, A, DATATYPE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights
B, A = int(os.environ.get("B", 32)), int(os.environ.get("A", 4))
NAMES = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 4: "head", 5: "fc4_dgrad", 6: "f4w", 7: "c3d", 8: "c3w", 9: "c2d", 10: "c2w", 11: "c1w", 12: "update", 16: "bwd3", 17: "bwd2", 18: "bwd1", 24: "wgrads"}
ws, wt = xavier_weights(A, 1), xavier_weights(A, 2)
mb = random_minibatch(B, A, 3, reward_range=(-2, 3))
args = make_args(batch_size=B, datatype=os.environ.get("DATATYPE", "float32"))


def run(spec):
    net = sd.DeepQNetwork(A, args)
    net.set_weights(wt, 1); net.set_weights(ws, 0)
    for kv in [x for x in spec.split(",") if x]:
        k, v = kv.split("="); net.set_option(k, int(v))
    net.set_option("keep_gradients", 1)
    net.train(mb)
    g = [net.get_layer(i, 3) for i in range(5)]
    q = net.last_q()[0]
    net.set_option("keep_gradients", 0)
    for _ in range(20):
        net.train(mb)
    net.profile(True, -1); net.profile_reset()
    for _ in range(200):
        net.train(mb)
    us = {p["id"]: p["total_ms"] / p["launches"] * 1e3 for p in net.profile_read() if p["launches"]}
    net.profile(False)
    return g, q, us


g0, q0, us0 = run("")
print("defaults: " + "  ".join("%s %.2f" % (NAMES[k], us0[k]) for k in sorted(us0) if k in NAMES), flush=True)
for spec in sys.argv[1:]:
    g, q, us = run(spec)
    gerr = max(float(np.abs(a - b).max() / max(1e-6, np.abs(b).max())) for a, b in zip(g, g0))
    print("%-20s grad rel %.1e q abs %.1e | " % (spec, gerr, float(np.abs(q - q0).max())) + "  ".join("%s %.2f" % (NAMES[k], us[k]) for k in sorted(us) if k in NAMES), flush=True)
