"""Experiment: why is the driver's 20-step timed region ~5 us/step slower than steady state?  Successive 20-step calls right after the
bench's own start-up sequence (1 M-frame ring upload, gather measurements, 5 warm-up steps)."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = 32, 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(int(os.environ.get("RING", 1000000)), args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
idx = np.array([mem.sample_indexes().copy() for _ in range(256)])
mem.bench_gather(idx, iters=512)
net.train_from_memory(mem, 5, mt_state=mt, want_cost=False); net.sync()
if os.environ.get("PRE_STEPS"):                  # sustained training before the short calls
    net.train_from_memory(mem, int(os.environ["PRE_STEPS"]), mt_state=mt, want_cost=False); net.sync()
if os.environ.get("GAP_MS"): time.sleep(float(os.environ["GAP_MS"]) / 1e3)
out = []
for k in range(12):
    t = time.perf_counter(); net.train_from_memory(mem, 20, mt_state=mt, want_cost=False); net.sync(); out.append((time.perf_counter() - t) * 1e6 / 20)
print("us/step of successive 20-step calls after 5 warm-up steps:", " ".join("%.1f" % x for x in out))
