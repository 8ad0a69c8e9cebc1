"""Tuning helper (not product): step rate with the XCD-contiguous tile map switched per launch / per problem
(sdqn_net_set_option "xcd:<kernel id>", value = problem mask + 1)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = int(os.environ.get("B", 32)), 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
N = int(os.environ.get("N", 3000))


def rate():
    net.train_from_memory(mem, 300, mt_state=mt, want_cost=False); net.sync()
    best = 0.0
    for _ in range(3):
        t = time.perf_counter(); net.train_from_memory(mem, N, mt_state=mt, want_cost=False); net.sync()
        best = max(best, N / (time.perf_counter() - t))
    return best


base = rate()
print("built-in %.0f steps/s" % base, flush=True)
names = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 5: "fc4_dgrad", 16: "bwd3[f4w,c3d,c3w]", 17: "bwd2[-,c2d,c2w]", 18: "bwd1[-,c1w]"}
masks = {0: [1], 1: [1], 2: [1], 3: [0, 1], 5: [1], 16: [1, 2, 4, 3, 5, 6, 7], 17: [2, 4, 6], 18: [2]}
for kid, nm in names.items():
    row = {}
    for m in masks[kid]:
        net.set_option("xcd:%d" % kid, m + 1)
        row[m] = "%+.1f%%" % ((rate() / base - 1) * 100)
    net.set_option("xcd:%d" % kid, 0)
    print("%-20s mask -> step rate vs built-in: %s" % (nm, row), flush=True)
print("built-in again %.0f steps/s" % rate())
