"""Soak run (not a benchmark): the fused loop for SECONDS seconds with target syncs every 2 500 updates, new frames added to the ring
while it trains (Agent-style: 4 adds per update in bursts), periodic checks that the cost and every weight stay finite and that the
device reports no hand-off time-out or error; prints one summary line."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
SECONDS = float(os.environ.get("SECONDS_", 60))
B, A = 32, 4
args = make_args(batch_size=B, datatype=os.environ.get("DATATYPE", "float32"))
mem = sd.ReplayMemory(200000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
rng = np.random.RandomState(0)
frames = rng.randint(0, 256, size=(64, 84, 84), dtype=np.uint8)
t0 = time.time(); steps = adds = checks = 0; costs = []
while time.time() - t0 < SECONDS:
    for _ in range(10):
        if net.train_iterations % 2500 < 250:
            net.update_target_network()
        costs.append(net.train_from_memory(mem, 250, mt_state=mt, want_cost=True)); steps += 250
    for i in range(1000):                                   # what an agent would have added meanwhile
        mem.add(int(rng.randint(A)), int(rng.randint(-1, 2)), frames[i % 64], bool(rng.rand() < 0.005)); adds += 1
    net.sync()
    w = net.get_weights(0)
    assert all(np.isfinite(x).all() for x in w) and np.isfinite(costs[-1]), "non-finite state after %d steps" % steps
    checks += 1
el = time.time() - t0
print("soak %s: %d train steps + %d adds in %.1f s (%.0f steps/s incl. adds and checks), %d finiteness checks, cost first/last %.4g / %.4g, max |W| %.3g"
      % (args.datatype, steps, adds, el, steps / el, checks, costs[0], costs[-1], max(float(np.abs(x).max()) for x in w)))
