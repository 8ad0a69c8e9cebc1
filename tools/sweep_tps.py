"""Tuning helper (not product): step rate against the split-K slab size (32-deep chunks per slab) of the three conv weight gradients,
sdqn_net_set_option "tps:<layer>".  B=256 [DATATYPE=float16] [CAND="{1:[..],2:[..],3:[..]}"] python tools/sweep_tps.py"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = int(os.environ.get("B", 256)), 4
DT = os.environ.get("DATATYPE", "float32")
args = make_args(batch_size=B, datatype=DT)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
N = int(os.environ.get("N", 600))
def rate():
    net.train_from_memory(mem, 100, mt_state=mt, want_cost=False); net.sync()
    r = []
    for _ in range(3):
        t = time.perf_counter(); net.train_from_memory(mem, N, mt_state=mt, want_cost=False); net.sync()
        r.append(N / (time.perf_counter() - t))
    return max(r)
PIX = {1: 400, 2: 81, 3: 49}
T = {l: -(-B * PIX[l] // 32) for l in PIX}
dflt = {1: -(-T[1] // 25), 2: min(T[2], 8), 3: -(-T[3] // 4)}                  # as sdqn_net_create picks them
if B >= 128: dflt = {1: min(T[1], 100 if DT == "float16" else 50), 2: min(T[2], 18), 3: min(T[3], 20)}
print("B", B, DT, "chunks", T, "default tps", dflt, "base", round(rate()), flush=True)
cand = eval(os.environ.get("CAND", "{}")) or {1: [50, 56, 80, 100], 2: [11, 14, 18, 24, 36], 3: [7, 10, 13, 16, 20, 25, 33]}
best = dict(dflt)
for l in (3, 2, 1):
    res = {}
    for v in cand[l]:
        net.set_option("tps:%d" % l, v); res[v] = round(rate())
    net.set_option("tps:%d" % l, dflt[l]); res["default %d" % dflt[l]] = round(rate())
    print("layer", l, "tps -> rate", res, flush=True)
    bv = max(cand[l], key=lambda v: res[v])
    if res[bv] > res["default %d" % dflt[l]] * 1.003: best[l] = bv
    net.set_option("tps:%d" % l, best[l])
print("best", best, "slabs", {l: -(-T[l] // best[l]) for l in best}, "rate", round(rate()))
