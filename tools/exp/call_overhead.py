"""Experiment: fixed cost of one train_from_memory call (first-step prep launch, enqueue start, final synchronise) at steady state."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = 32, 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
net.train_from_memory(mem, 2000, mt_state=mt, want_cost=False); net.sync()
res = {}
for n in (1, 2, 5, 20, 100, 1000):
    reps = max(5, 4000 // n)
    t = time.perf_counter()
    for _ in range(reps):
        net.train_from_memory(mem, n, mt_state=mt, want_cost=False); net.sync()
    dt = (time.perf_counter() - t) / reps
    res[n] = dt * 1e6
    print("steps per call %5d: %8.1f us per call, %7.2f us per step" % (n, dt * 1e6, dt * 1e6 / n), flush=True)
per = (res[1000] - res[100]) / 900
print("marginal step %.2f us; fixed cost per call: %s" % (per, {n: round(res[n] - n * per, 1) for n in res}))
t = time.perf_counter()
for _ in range(2000): net.sync()
print("net.sync() on an idle stream: %.2f us" % ((time.perf_counter() - t) / 2000 * 1e6))
