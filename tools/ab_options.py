"""Tuning helper (not product): end-to-end step rate (no event brackets) with sdqn_net_set_option switches.
OPTS='[[("nw:1", 8)], [("xcd:16", 3)]]' python tools/ab_options.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = int(os.environ.get("B", 32)), 4
args = make_args(batch_size=B, datatype=os.environ.get("DATATYPE", "float32"))
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
def rate(N=int(os.environ.get("N", 6000))):
    net.train_from_memory(mem, 300, mt_state=mt, want_cost=False); net.sync()
    r = []
    for _ in range(3):
        t = time.perf_counter(); net.train_from_memory(mem, N, mt_state=mt, want_cost=False); net.sync()
        r.append(N / (time.perf_counter() - t))
    return max(r)
print("base", round(rate()))
OPTS = eval(os.environ.get("OPTS", "[]")) or ([("f4_share3", 70), ("f4_share2", 15)], [("xcd_map", 1)])
for opts in OPTS:
    for k, v in opts: net.set_option(k, v)
    print(opts, round(rate()))
    for k, v in opts:
        if not k.startswith("tps:") and k != "s4": net.set_option(k, 100 if k == "f4_share3" else 0)
print("base", round(rate()))
