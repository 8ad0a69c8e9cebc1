"""Experiment (not product): s_memtime stamps of the sample-stationary forward convolutions (csrc/conv_ss.h), -DSDQN_TIMING build.
Slots of workgroup b: matrix wave 0: 0 entry, 1 barrier #0 passed (first rows + chunk 0 in LDS), 2 / 3 after chunks 0 / 4, 4 K-outer phase
done, 5 final phase done; staging wave (thread 256): 6 prologue committed, 7 last output store issued.     python tools/exp/ss_stamps.py"""
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
A = 3
MAXB = 1024
for B in ([int(x) for x in sys.argv[1:]] or [256]):
    args = make_args(batch_size=B)
    mem = sd.ReplayMemory(20000, args); fill_ring(mem, 1, A)
    mt = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt, 5)
    random.seed(1)
    for kid, name in ((1, "conv2_fwd"), (2, "conv3_fwd")):
        net = sd.DeepQNetwork(A, args); net.update_target_network()
        net.train_from_memory(mem, 5, mt_state=mt, want_cost=False); net.sync()
        idx = np.array(mem.sample_indexes())
        out = np.zeros((MAXB, 8), np.uint64)
        L.check(lib.sdqn_debug_time_kernel(net._h, mem._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), kid, out.ctypes.data_as(C.POINTER(C.c_uint64)), MAXB))
        v = out[out[:, 0] > 0].astype(np.int64)
        if len(v) == 0:
            print(name, "no stamps"); continue
        rel = v - v[:, :1]
        names = ["barrier0", "chunk0", "chunk4", "kouter", "final", "stg_prologue", "stg_end"]
        print("B=%d %-9s blocks %d; cycles after matrix-wave entry, p10 / median / p90:" % (B, name, len(v)))
        for c in range(1, 8):
            x = rel[:, c]
            print("   %-13s %7d %7d %7d" % (names[c - 1], np.percentile(x, 10), np.median(x), np.percentile(x, 90)))
        t0 = v[:, 0].min()
        print("   launch span (first entry -> last stamp): %d cycles; entries spread %d" % (v.max() - t0, v[:, 0].max() - t0), flush=True)
        del net
