"""Experiment helper: float16 mode at B >= 128, half block-tile routine vs the wave-tile routines (option bt = 0), per launch + rate + parity."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights
from bench import fill_ring
B, A = int(os.environ.get("B", 256)), 3
NAMES = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 4: "head", 5: "fc4_dgrad", 7: "conv3_dgrad", 9: "conv2_dgrad", 12: "update", 16: "bwd3", 17: "bwd2", 18: "bwd1", 24: "wgrads"}
ws, wt = xavier_weights(A, 1), xavier_weights(A, 2)
mb = random_minibatch(B, A, 3, reward_range=(-2, 3))
args = make_args(batch_size=B, datatype="float16")
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
ref = None
for spec in sys.argv[1:] or ["bt=0", ""]:
    net = sd.DeepQNetwork(A, args); net.set_weights(wt, 1); net.set_weights(ws, 0)
    for kv in [x for x in spec.split(",") if x]:
        k, v = kv.split("="); net.set_option(k, int(v))
    net.set_option("keep_gradients", 1); net.train(mb)
    g = [net.get_layer(i, 3) for i in range(5)]; q = net.last_q()[0]
    net.set_option("keep_gradients", 0)
    for _ in range(5): net.train(mb)
    net.profile(True, -1); net.profile_reset()
    for _ in range(30): net.train(mb)
    us = {p["id"]: p["total_ms"] / p["launches"] * 1e3 for p in net.profile_read() if p["launches"]}
    net.profile(False)
    mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
    net.train_from_memory(mem, 100, mt_state=mt, want_cost=False); net.sync()
    t = time.perf_counter(); net.train_from_memory(mem, 400, mt_state=mt, want_cost=False); net.sync(); rate = 400 / (time.perf_counter() - t)
    if ref is None: ref = (g, q)
    gerr = [float(np.linalg.norm((a - b).ravel()) / max(1e-12, np.linalg.norm(b.ravel()))) for a, b in zip(g, ref[0])]
    print("%-28s %5d steps/s (%.1f us) | %s | grad rel-Frobenius vs first %s q %.1e" % (
        spec or "(defaults)", rate, 1e6 / rate, "  ".join("%s %.1f" % (NAMES[k], us[k]) for k in sorted(us) if k in NAMES), ["%.1e" % e for e in gerr], float(np.abs(q - ref[1]).max())), flush=True)
