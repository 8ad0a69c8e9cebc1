"""Experiment helper: step rate of an alternative build of the library (LIB=path/to/libsdqn_hip_xxx.so), same loop as tools/ab_options.py."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd._lib as L
if os.environ.get("LIB"):
    L.lib_path = lambda: os.path.join(ROOT, os.environ["LIB"])
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = int(os.environ.get("B", 32)), 4
args = make_args(batch_size=B, datatype=os.environ.get("DATATYPE", "float32"))
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
N = int(os.environ.get("N", 6000))
net.train_from_memory(mem, 300, mt_state=mt, want_cost=False); net.sync()
r = []
for _ in range(4):
    t = time.perf_counter(); net.train_from_memory(mem, N, mt_state=mt, want_cost=False); net.sync(); r.append(N / (time.perf_counter() - t))
print(os.environ.get("LIB", "default"), "B", B, os.environ.get("DATATYPE", "float32"), "steps/s:", " ".join("%.0f" % x for x in r))
