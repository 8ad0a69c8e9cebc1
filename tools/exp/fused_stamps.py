"""Experiment (not product): s_memtime stamps of the fused update + conv1 launch (timing build)."""
import ctypes as C, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
mem = sd.ReplayMemory(200000, args); fill_ring(mem, 1, 4)
net = sd.DeepQNetwork(4, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt, 5)
net.train_from_memory(mem, 20, mt_state=mt, want_cost=False); net.sync()
random.seed(1)
for kid, nm in ((104, "update alone"), (100, "conv1 alone (warm)"), (103, "fused update+conv1")):
    idx = np.array(mem.sample_indexes())
    out = np.zeros((4096, 8), np.uint64)
    L.check(lib.sdqn_debug_time_kernel(net._h, mem._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), kid, out.ctypes.data_as(C.POINTER(C.c_uint64)), 4096))
    v = out.astype(np.int64)
    live = v[:, 0] > 0
    t0 = v[live, 0].min()
    end = np.maximum(v[:, 6], v[:, 7])
    print("%s: %d blocks stamped, kernel span (first entry -> last exit) %d cycles" % (nm, live.sum(), end[live].max() - t0))
    if kid == 103:
        nupd = 608 + 16 + 2
        for name, sl in (("update blocks", slice(0, nupd)), ("W1 blocks", slice(0, 64)), ("target conv1", slice(nupd, nupd + 100)), ("online conv1", slice(nupd + 100, nupd + 200))):
            w = v[sl]; ok = w[:, 0] > 0
            e = np.maximum(w[ok, 6], w[ok, 7])
            print("   %-14s n=%3d  entry: first %6d median %6d last %6d | exit: median %6d last %6d | life median %6d" % (
                name, ok.sum(), w[ok, 0].min() - t0, np.median(w[ok, 0]) - t0, w[ok, 0].max() - t0, np.median(e) - t0, e.max() - t0, np.median(e - w[ok, 0])))
        w = v[nupd + 100:nupd + 200]
        print("   online conv1 phases (median, from entry): frame loads issued %d, plane fill done %d, barrier %d, mfma done %d, end %d" % tuple(
            int(np.median(w[:, c] - w[:, 0])) for c in (1, 2, 3, 5, 6)))
        w = v[nupd:nupd + 100]
        print("   target conv1 phases (median, from entry): frame loads issued %d, plane fill done %d, barrier %d, mfma done %d, end %d" % tuple(
            int(np.median(w[:, c] - w[:, 0])) for c in (1, 2, 3, 5, 6)))
