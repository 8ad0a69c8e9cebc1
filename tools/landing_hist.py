"""Experiment (not product; `make -C simple_dqn_amd/csrc timing` build): WHEN do the operands of a tile land?  Per-wave s_memtime stamps of the
latency engine's tile routine (gemm_engine.h: SDQN_WSTAMP) around ONE launch id inside a real B = 32 train step (armed and disarmed
stream-ordered right around the launch: the stamped launch finds the caches as its predecessors in the step left them).  Per launch id:
percentiles over all waves of (a) issuing the chunk's loads, (b) issue -> A operand landed, (c) A landed -> B landed (in-order return: B's
loads were issued behind A's), (d) the whole wait, and the same per XCC.  VERDICT r4 item 4.   env: REPS (stamped steps per id, default 6)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SDQN_LIB_PATH"] = os.path.join(ROOT, "simple_dqn_amd", "libsdqn_hip_timing.so")
import numpy as np
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
lib = sd.load()
lib.sdqn_debug_time_step_waves.restype = C.c_int
lib.sdqn_debug_time_step_waves.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
B, A, REPS = 32, 4, int(os.environ.get("REPS", 6))
args = make_args(batch_size=B)
mem = sd.ReplayMemory(int(os.environ.get("RING", 100000)), args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt, 5)
net.train_from_memory(mem, 300, mt_state=mt, want_cost=False); net.sync()
MAXB = 1024
# launch id -> (name, [(problem, first block, blocks, waves per block)])   (block ranges of the multi-problem launches: sdqn_kernels.hip)
IDS = {1: ("conv2_fwd", [("conv2_fwd (A: a1 patches, gathered 16 B per lane; B: W2, row-major)", 0, 324, 16)]),
       3: ("fc4_fwd", [("fc4_fwd (A: a3 rows staged through LDS; B: W4 k-slab)", 0, 224, 14)]),
       5: ("fc4_dgrad", [("fc4_dgrad (A: delta4 staged; B: W4^T panel)", 0, 98, 16)]),
       16: ("bwd3", [("conv3_dgrad", 0, 162, 8), ("conv3_wgrad", 162, 144, 8)]),
       17: ("bwd2", [("conv2_dgrad", 0, 400, 8), ("conv2_wgrad", 400, 352, 8)])}


def pct(x, ps=(10, 50, 90, 99)):
    return " / ".join("%5d" % int(np.percentile(x, p)) for p in ps) if len(x) else "-"


print("# tools/landing_hist.py: B = 32, A = 4, float32, ring path; timing build (stamps cost ~10 %% of a wave's life); cycles = s_memtime ticks (core clock)")
print("# percentiles 10 / 50 / 90 / 99 over all stamped waves of %d stamped steps per launch id" % REPS)
for kid, (name, probs) in IDS.items():
    W, X = [], []
    for rep in range(REPS):
        w = np.zeros((MAXB, 16, 4), np.uint64); x = np.zeros(MAXB, np.uint64)
        rc = lib.sdqn_debug_time_step_waves(net._h, mem._h, mt, kid, 20, w.ctypes.data_as(C.POINTER(C.c_uint64)), x.ctypes.data_as(C.POINTER(C.c_uint64)), MAXB)
        assert rc == 0, lib.sdqn_last_error()
        W.append(w.astype(np.int64)); X.append(x.astype(np.int64))
    for pname, first, nb, nw in probs:
        iss, la, lb, tot, skew, xcc = [], [], [], [], [], []
        for w, x in zip(W, X):
            blk = w[first:first + nb, :nw, :]
            ok = (blk[:, :, 0] > 0) & (blk[:, :, 3] > 0)
            if not ok.any():
                continue
            xb = np.repeat(x[first:first + nb, None], nw, axis=1)
            t0 = np.zeros_like(blk[:, :, 0])
            for q in range(8):                                   # (every XCC has its own s_memtime origin)
                sel = ok & (xb == q)
                if sel.any():
                    t0[xb == q] = blk[:, :, 0][sel].min()
            iss.append((blk[:, :, 1] - blk[:, :, 0])[ok]); la.append((blk[:, :, 2] - blk[:, :, 1])[ok]); lb.append((blk[:, :, 3] - blk[:, :, 2])[ok])
            tot.append((blk[:, :, 3] - blk[:, :, 0])[ok]); skew.append((blk[:, :, 0] - t0)[ok])
            xcc.append(np.repeat(x[first:first + nb, None], nw, axis=1)[ok])
        if not iss:
            print("%-10s %s: no stamps (this launch does not run the latency engine's tile routine)" % (name, pname)); continue
        iss, la, lb, tot, skew, xcc = map(np.concatenate, (iss, la, lb, tot, skew, xcc))
        print("%-10s %s: %d wave samples" % (name, pname, len(tot)))
        print("    wave start after its XCC's first wave        %s" % pct(skew))
        print("    issuing the chunk's loads                  %s" % pct(iss))
        print("    all issued -> A landed                     %s" % pct(la))
        print("    A landed -> B landed                       %s" % pct(lb))
        print("    first load issued -> operands landed       %s   (MFMA work of a chunk: 1 024 cycles)" % pct(tot))
        print("    the last, per XCC (median / p90):          " + "  ".join("%d: %d / %d" % (q, np.median(tot[xcc == q]), np.percentile(tot[xcc == q], 90)) for q in range(8) if (xcc == q).any()))
