"""Experiment (round 6, with tools/exp/stagedx_wgrad.patch applied): the conv2 / conv3 weight gradients' operands through LDS (StagedX, option wt bit 9) against direct dword loads at
B = 32: bit-identity of the gradients, per-launch times, alternating step rates.   python tools/exp/stagedx_ab.py"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from bench import fill_ring
from oracle.dqn_numpy import xavier_weights
B, A = 32, 4
mb = random_minibatch(B, A, 11, reward_range=(-2, 3))
nets = {}
for tag, wt in (("stagedx", 1023), ("direct", 511)):
    n = sd.DeepQNetwork(A, make_args(batch_size=B)); n.set_weights(xavier_weights(A, 2), 1); n.set_weights(xavier_weights(A, 1), 0)
    n.set_option("keep_gradients", 1); n.set_option("wt", wt); n.train(mb); nets[tag] = n
for i in range(5):
    print("layer %d gradient bit-identical: %s" % (i, np.array_equal(nets["stagedx"].get_layer(i, 3), nets["direct"].get_layer(i, 3))))
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
for wt in (1023, 511, 1023, 511):
    net.set_option("wt", wt)
    net.train_from_memory(mem, 300, mt_state=mt, want_cost=False)
    net.profile(True, -1); net.profile_reset()
    net.train_from_memory(mem, 200, mt_state=mt, want_cost=False)
    prof = {p["name"].split("(")[0]: p["total_ms"] / p["launches"] * 1e3 for p in net.profile_read() if p["launches"] >= 200}
    net.profile(False); net.sync()
    r = []
    for _ in range(3):
        t = time.perf_counter(); net.train_from_memory(mem, 3000, mt_state=mt, want_cost=False); net.sync(); r.append(3000 / (time.perf_counter() - t))
    print("wt %4d: %d steps/s | bwd3 %.2f bwd2 %.2f bwd1 %.2f us" % (wt, max(r), prof.get("bwd3", 0), prof.get("bwd2", 0), prof.get("bwd1", 0)), flush=True)
