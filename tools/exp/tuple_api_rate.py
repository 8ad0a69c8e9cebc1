"""Experiment (not product): step rate of the reference-compatible tuple API — mem.getMinibatch() -> net.train(minibatch) — the
loop body of the reference's Agent.train (src/agent.py:108-114), with the library's ReplayMemory (pinned buffers) and with
foreign pageable arrays."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd._lib as _L
if os.environ.get("SDQN_LIB"): _L.lib_path = lambda: os.path.join(os.path.dirname(os.path.abspath(_L.__file__)), os.environ["SDQN_LIB"])   # A/B of two builds
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from bench import fill_ring
B, A = 32, 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
random.seed(1)
for _ in range(50): net.train(mem.getMinibatch())
net.sync(); t = time.perf_counter(); N = 2000
for _ in range(N): net.train(mem.getMinibatch())
net.sync(); print("getMinibatch + train(tuple): %.0f steps/s" % (N / (time.perf_counter() - t)))
print("served as (calls, states in place, nothing uploaded):", net.tuple_counters())
if os.environ.get("ONLY_TUPLE"): sys.exit(0)
mb = random_minibatch(B, A, 3)
for _ in range(50): net.train(mb)
net.sync(); t = time.perf_counter()
for _ in range(N): net.train(mb)
net.sync(); print("train(pageable tuple) only: %.0f steps/s" % (N / (time.perf_counter() - t)))
