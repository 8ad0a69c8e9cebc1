"""Experiment: throughput-regime tile routine (set_option big_tiles) at B >= 128 — parity vs the oracle + step rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import simple_dqn_amd as sd
from oracle.dqn_numpy import OracleDQN, xavier_weights
from util import make_args, random_minibatch
from bench import fill_ring
B, A = int(os.environ.get("B", 256)), 3
args = make_args(batch_size=B)
ws, wt = xavier_weights(A, 3), xavier_weights(A, 4)
o = OracleDQN(A, batch_size=B, weights=ws); o.Wt = [w.copy() for w in wt]
mb = random_minibatch(B, A, 5)
g, cost, deltas, preq = o.gradients(mb)
for big in (0, 1):
    net = sd.DeepQNetwork(A, args); net.set_weights(wt, 1); net.set_weights(ws, 0)
    net.set_option("big_tiles", big); net.set_option("keep_gradients", 1)
    q = net.predict(mb[0]); print("big", big, "predict max err %.2e" % np.abs(q - o.predict(mb[0])).max())
    net.train(mb)
    for i in range(5):
        gi = net.get_layer(i, 3); print("   grad layer %d rel err %.2e" % (i, np.abs(gi - g[i]).max() / max(1e-3, np.abs(g[i]).max())))
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
for big in (0, 1, 0, 1):
    net = sd.DeepQNetwork(A, args); net.update_target_network(); net.set_option("big_tiles", big)
    mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
    net.train_from_memory(mem, 100, mt_state=mt, want_cost=False); net.sync()
    r = []
    for _ in range(3):
        t = time.perf_counter(); net.train_from_memory(mem, 600, mt_state=mt, want_cost=False); net.sync(); r.append(600 / (time.perf_counter() - t))
    print("big", big, "B", B, "steps/s %.0f (%.1f us)" % (max(r), 1e6 / max(r)))
    net.set_option("fused_launches", 0); net.profile(True, -1); net.profile_reset()
    net.train_from_memory(mem, 100, mt_state=mt, want_cost=False)
    print("    ", {p["name"][:12]: round(p["total_ms"] / p["launches"] * 1e3, 1) for p in net.profile_read() if p["launches"]})
    net.profile(False)
