import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd._lib as _L
if os.environ.get("SDQN_LIB"): _L.lib_path = lambda: os.path.join(os.path.dirname(os.path.abspath(_L.__file__)), os.environ["SDQN_LIB"])   # A/B of two builds
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = 32, 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
net.train_from_memory(mem, 300, mt_state=mt, want_cost=False); net.sync()
r = []
for _ in range(int(os.environ.get("REPS", 5))):
    t = time.perf_counter(); net.train_from_memory(mem, int(os.environ.get("STEPS", 6000)), mt_state=mt, want_cost=False); net.sync()
    r.append(int(os.environ.get("STEPS", 6000)) / (time.perf_counter() - t))
print(os.path.basename(ROOT) or "current", [round(x) for x in r], "max", round(max(r)))
