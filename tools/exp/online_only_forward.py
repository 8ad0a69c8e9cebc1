"""Experiment (not product): how much of the forward launches is the target net?  HIP-event time per launch of the
forward stages with both nets (train step) vs the online net only (predict path, nz = 1) at B = 32: the upper bound of what
hoisting the target forward of step i+1 into step i's backward launches could take off the forward chain."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from bench import fill_ring
B, A = 32, 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(50000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
st = random_minibatch(B, A, 1)[0]
def prof(fn, n):
    fn(20); net.sync()
    net.profile(True, -1); net.profile_reset(); fn(n)
    r = {p["name"].split("(")[0]: round(p["total_ms"] / p["launches"] * 1e3, 2) for p in net.profile_read() if p["launches"] >= n}
    net.profile(False)
    return r
both = prof(lambda n: net.train_from_memory(mem, n, mt_state=mt, want_cost=False), 300)
def pred(n):
    for _ in range(n): net.predict(st)
online = prof(pred, 300)
print("train step (both nets):", both)
print("predict (online net only):", online)
f = ["conv1_fwd", "conv2_fwd", "conv3_fwd", "fc4_fwd", "head"]
print("forward chain both nets %.1f us, online only %.1f us, difference %.1f us (event-bracketed launches)"
      % (sum(both[k] for k in f), sum(online[k] for k in f), sum(both[k] - online[k] for k in f)))
