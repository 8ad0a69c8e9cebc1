"""Tuning helper (not product): per-kernel HIP-event time for different waves-per-tile settings."""
import sys, os, random, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
import ctypes as C
from simple_dqn_amd import _lib
B, A = int(os.environ.get("B", 32)), 4
args = make_args(batch_size=B, datatype=os.environ.get("DATATYPE", "float32"))
mem = sd.ReplayMemory(50000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
net.set_option("fused_launches", 0)
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
def measure(n=300):
    net.train_from_memory(mem, 50, mt_state=mt, want_cost=False)
    net.profile(True, -1); net.profile_reset()
    net.train_from_memory(mem, n, mt_state=mt, want_cost=False)
    r = {p["id"]: p["total_ms"] / p["launches"] * 1e3 for p in net.profile_read() if p["launches"]}
    net.profile(False)
    return r
base = measure()
print("base", {k: round(v, 2) for k, v in base.items()})
names = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 5: "fc4_dgrad", 6: "fc4_wgrad", 7: "conv3_dgrad", 8: "conv3_wgrad", 9: "conv2_dgrad", 10: "conv2_wgrad", 11: "conv1_wgrad"}
for kid, nm in names.items():
    row = {}
    for nw in (1, 2, 4, 8, 16):
        net.set_option("nw:%d" % kid, nw)
        try:
            row[nw] = round(measure(200)[kid], 2)
        except Exception as e:
            row[nw] = str(e)[:40]
    net.set_option("nw:%d" % kid, 0)
    print(nm, "builtin %.2f" % base[kid], row, flush=True)
