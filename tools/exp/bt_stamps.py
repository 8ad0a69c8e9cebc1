"""Experiment (not product): s_memtime stamps of the block-tile (bt_tile) and ping-pong (pp_tile) routines at B = 256, -DSDQN_TIMING build.
Slots of workgroup b (wave 0, thread 0): 0 entry, 1 prologue done (chunk 0 staged, fragments read), 2 / 3 / 4 after chunks 3 / 7 / 11,
5 main loop done, 6 partial sums combined (pp), 7 epilogue stores issued.     python tools/exp/bt_stamps.py"""
import ctypes as C, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd._lib as L
L.lib_path = lambda: os.path.join(os.path.dirname(os.path.abspath(L.__file__)), "libsdqn_hip_timing.so")
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
lib = sd.load()
lib.sdqn_debug_time_kernel.restype = C.c_int
lib.sdqn_debug_time_kernel.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_uint64), C.c_int]
B, A = 256, 3
args = make_args(batch_size=B)
mem = sd.ReplayMemory(50000, args); fill_ring(mem, 1, A)
mt = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt, 5)
random.seed(1)
MAXB = 4096
for kid, name, specs in ((5, "fc4_dgrad", ("bt:5=10", "bt:5=11")), (2, "conv3_fwd", ("bt:2=10",)), (1, "conv2_fwd", ("bt:1=10",))):
    for spec in specs:
        net = sd.DeepQNetwork(A, args); net.update_target_network()
        net.set_option("fused_launches", 0)
        k, v = spec.split("="); net.set_option(k, int(v))
        net.train_from_memory(mem, 5, mt_state=mt, want_cost=False); net.sync()
        idx = np.array(mem.sample_indexes())
        out = np.zeros((MAXB, 8), np.uint64)
        L.check(lib.sdqn_debug_time_kernel(net._h, mem._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), kid, out.ctypes.data_as(C.POINTER(C.c_uint64)), MAXB))
        v = out[out[:, 0] > 0].astype(np.int64)
        if len(v) == 0:
            print(name, spec, "no stamps"); continue
        t0 = v[:, 0].min()
        rel = v - v[:, :1]
        med = [int(np.median(rel[rel[:, c] > 0, c])) if (rel[:, c] > 0).any() else -1 for c in range(1, 8)]
        end = v.max(axis=1)
        print("%-10s %-9s blocks %4d | median cycles after entry: ph8 start %d, +store/load issued %d, +mfma retired %d, ph9 start %d, +frags landed/stores %d, ph10 start %d, end %d" % (name, spec, len(v), *med), flush=True)
        del net
