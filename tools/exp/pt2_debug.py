import ctypes as C, os, sys, random
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd._lib as L
L.lib_path = lambda: os.path.join(ROOT, "simple_dqn_amd", "libsdqn_hip_timing.so")
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
lib = sd.load()
lib.sdqn_debug_time_kernel.restype = C.c_int
lib.sdqn_debug_time_kernel.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_uint64), C.c_int]
args = make_args(batch_size=32)
mem = sd.ReplayMemory(20000, args); fill_ring(mem, 1, 4)
net = sd.DeepQNetwork(4, args); net.update_target_network()
print("train with timing lib...", flush=True)
mt = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt, 5)
net.train_from_memory(mem, 3, mt_state=mt, want_cost=False); net.sync()
print("ok", flush=True)
random.seed(1); idx = np.array(mem.sample_indexes())
for kid in (102, 2, 100):
    out = np.zeros((4096, 8), np.uint64)
    print("kid", kid, flush=True)
    L.check(lib.sdqn_debug_time_kernel(net._h, mem._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), kid, out.ctypes.data_as(C.POINTER(C.c_uint64)), 4096))
    print("done", int((out[:, 0] > 0).sum()), flush=True)
