"""[historical, rounds 2-4: the float32 register-blocked routine and its option "rb:<id>" left the library in round 5 — runs on the
tree of tools/exp/experiments_r04.patch]  Bring-up helper (not product): s_memtime phase stamps + placement (HW_ID / XCC_ID) of the register-blocked routine at
B = 256, from the -DSDQN_TIMING build.   python tools/rb_stamps.py "<kernel id>:<menu>" ..."""
import ctypes as C, os, sys, collections
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
mem = sd.ReplayMemory(30000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
net.set_option("fused_launches", 0)
mt = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt, 5)
net.train_from_memory(mem, 3, mt_state=mt, want_cost=False); net.sync()
import random
random.seed(1); idx = np.array(mem.sample_indexes())
MAXB = 1 << 15
for spec in sys.argv[1:]:
    kid, menu = [int(x) for x in spec.split(":")]
    net.set_option("rb:%d" % kid, menu)
    out = np.zeros((MAXB, 8), np.uint64)
    L.check(lib.sdqn_debug_time_kernel(net._h, mem._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), kid, out.ctypes.data_as(C.POINTER(C.c_uint64)), MAXB))
    v = out[out[:, 0] > 0]
    t = v[:, :5].astype(np.int64)
    t0 = t[:, 0].min()
    span = int(t[:, 1:5].max() - t0)
    hw = v[:, 7]
    xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
    hwid = hw & np.uint64(0xFFFFFFFF)
    cu = (hwid >> np.uint64(8)) & np.uint64(0xF); sh = (hwid >> np.uint64(12)) & np.uint64(1); se = (hwid >> np.uint64(13)) & np.uint64(7)
    place = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    cnt = np.array(sorted(place.values()))
    d = np.diff(t, axis=1)
    print("kernel %d menu %d: blocks %d  span %d cyc | start spread %d | median [entry->loads issued %d, ->first chunk done %d, ->loop done %d, ->epilogue done %d] block life median %d max %d"
          % (kid, menu, len(v), span, int(t[:, 0].max() - t0), *[int(np.median(d[:, i])) for i in range(4)], int(np.median(t[:, 4] - t[:, 0])), int((t[:, 4] - t[:, 0]).max())))
    print("    placement: %d distinct (xcc,se,sh,cu) used; workgroups per used CU min %d median %d max %d; per XCC %s"
          % (len(place), cnt.min(), int(np.median(cnt)), cnt.max(), sorted(collections.Counter(xcc.tolist()).items())))
    net.set_option("rb:%d" % kid, 0)
