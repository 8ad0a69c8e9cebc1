"""Experiment helper (not product): the block-tile engine's arithmetic modes (option bt_x: 0 fp32 MFMA, 9 / 6 exact bf16x3 splits)
— per-launch times, gradients against the fp32-MFMA result, step rate.   python tools/exp/bt_x3.py [B] [A] [extra opts k=v,...]"""
import os, sys, time, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights
from bench import fill_ring
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = int(sys.argv[2]) if len(sys.argv) > 2 else 3
EXTRA = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] else [])]
NAMES = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 4: "head", 5: "fc4_dgrad", 12: "update", 16: "bwd3", 17: "bwd2", 18: "bwd1"}
ws, wt = xavier_weights(A, 1), xavier_weights(A, 2)
mb = random_minibatch(B, A, 3, reward_range=(-2, 3))
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
ref = None
for x in (0, 9, 6):
    net = sd.DeepQNetwork(A, args); net.set_weights(wt, 1); net.set_weights(ws, 0)
    for k, v in EXTRA + [("bt_x", x)]:
        net.set_option(k, v)
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
    gerr = [float(np.abs(a - b).max() / max(1e-9, np.abs(b).max())) for a, b in zip(g, ref[0])]
    print("bt_x=%d  %5d steps/s (%.1f us) | %s | grad rel-to-max err vs fp32-MFMA %s  q %.1e" % (
        x, rate, 1e6 / rate, "  ".join("%s %.1f" % (NAMES[k], us[k]) for k in sorted(us) if k in NAMES), ["%.1e" % e for e in gerr], float(np.abs(q - ref[1]).max())), flush=True)
