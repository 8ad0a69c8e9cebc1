"""Experiment (not product): per-phase s_memtime stamps of the tile engine, from the -DSDQN_TIMING build."""
import ctypes as C, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd._lib as L
L.lib_path = lambda: os.path.join(os.path.dirname(os.path.abspath(L.__file__)), "libsdqn_hip_timing.so")
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
lib = sd.load()
lib.sdqn_debug_time_kernel.restype = C.c_int
lib.sdqn_debug_time_kernel.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_uint64), C.c_int]
B, A = int(os.environ.get("B", 32)), 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(int(os.environ.get("RING", 50000)), args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
net.set_option("fused_launches", 0)
if os.environ.get("XCD"): net.set_option("xcd_map", int(os.environ["XCD"]))
mt = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt, 5)
net.train_from_memory(mem, 20, mt_state=mt, want_cost=False); net.sync()
random.seed(1); idx = np.array(mem.sample_indexes())
idx_cold = np.array(mem.sample_indexes())
names = {100: "conv1_bf16 warm", 101: "conv1_bf16 cold", 0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 5: "fc4_dgrad", 6: "fc4_wgrad", 7: "conv3_dgrad", 8: "conv3_wgrad", 9: "conv2_dgrad", 10: "conv2_wgrad", 11: "conv1_wgrad"}
MAXB = 4096
out = np.zeros((MAXB, 8), np.uint64)
L.check(lib.sdqn_debug_time_kernel(net._h, mem._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), 4, out.ctypes.data_as(C.POINTER(C.c_uint64)), MAXB))
v = out[out[:, 0] > 0].astype(np.int64)
d = np.diff(v[:, :8], axis=1)
print("head: blocks %d; median cycles per phase [entry->loads landed, ->shuffles done, ->barrier1, ->q/barrier2, ->thread0 TD, ->barrier3, ->stores]: %s; block life median %d"
      % (len(v), np.median(d, axis=0).astype(int).tolist(), int(np.median(v[:, 7] - v[:, 0]))), flush=True)
for kid, nm in names.items():
    out = np.zeros((MAXB, 8), np.uint64)
    use = idx_cold if kid == 101 else idx
    L.check(lib.sdqn_debug_time_kernel(net._h, mem._h, use.ctypes.data_as(C.POINTER(C.c_int64)), kid, out.ctypes.data_as(C.POINTER(C.c_uint64)), MAXB))
    v = out[out[:, 0] > 0].astype(np.int64)
    if len(v) == 0:
        print(nm, "no stamps"); continue
    t0 = v[:, 0].min()
    ph = ["entry->addr", "addr->loads issued", "issued->operands landed", "->mfma done", "->barrier", "->stores issued"]
    if kid >= 100: ph = ["entry->frame loads issued", "->plane loads+LDS stores issued", "->barrier passed", "->frame bytes landed", "->mfma done", "->stores issued"]
    cols = [1, 2, 3, 4, 5, 6]
    d = {}
    prev = v[:, 0]
    for name, c in zip(ph, cols):
        ok = v[:, c] > 0
        d[name] = int(np.median((v[ok, c] - prev[ok]))) if ok.any() else -1
        prev = np.where(ok, v[:, c], prev)
    end = v[:, 1:].max(axis=1)
    print("%-12s blocks %4d  start spread %6d cyc  block life median %6d  max %6d  kernel span %6d cyc | %s" % (
        nm, len(v), int(v[:, 0].max() - t0), int(np.median(end - v[:, 0])), int((end - v[:, 0]).max()), int(end.max() - t0), d), flush=True)
