"""Experiment helper: per-phase timeline of the one-launch acting forward (sdqn_act.hip) from its clock64 stamps, and call latencies.
   python tools/exp/act_stamps.py         prints, per XCC, when each phase's first item started / last item signalled (us after the
   earliest workgroup start of the launch; clock64 = s_memtime at 100 MHz), and predict_state latencies with act_kernel 1 / 0."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from simple_dqn_amd import _lib
from util import make_args
A = int(os.environ.get("A", 4))
args = make_args(batch_size=32)
net = sd.DeepQNetwork(A, args); net.update_target_network()
buf = sd.DeviceStateBuffer(args)
rng = np.random.RandomState(1)
for _ in range(6): buf.add(rng.randint(0, 256, (84, 84), dtype=np.uint8))
lib = sd.load()
q = np.empty(A, np.float32); st = np.zeros((256, 80), np.uint64)
GHZ = float(os.environ.get("GHZ", 2.4))                  # clock64 counts shader-engine cycles (not synchronised between XCCs / SEs)
for rep in range(4):
    _lib.check(lib.sdqn_net_debug_act(net._h, buf._h, q.ctypes.data_as(C.POINTER(C.c_float)), st.ctypes.data_as(C.POINTER(C.c_uint64))))
kinds = st[:, 0:78:2].astype(np.int64); clk = st[:, 1:79:2].astype(np.int64); xcc = st[:, 79].astype(np.int64)
code, item = kinds >> 16, kinds & 0xFFFF
rel = (clk - clk[:, :1]) / (GHZ * 1e3)                   # us after the workgroup's OWN first stamp
valid = np.arange(39)[None, :] < (kinds != 0).sum(1)[:, None] + 1
print("workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
names = {1: "conv1 item start", 11: "conv1 operands landed", 12: "conv1 MFMA done", 2: "conv1 signalled", 3: "conv2 item start (conv1 complete)", 13: "conv2 operands landed",
         14: "conv2 MFMA done", 4: "conv2 signalled", 5: "conv3 item start (conv2 complete)", 15: "conv3 operands landed", 16: "conv3 MFMA done", 6: "conv3 signalled",
         7: "fc4 chunk claimed (stripe known)", 8: "fc4: conv3 complete, W4 landed", 9: "fc4 chunk signalled", 10: "stripe Q partial written"}
print("time after the workgroup's own start, us: min / median / max over workgroups (all XCCs)")
for c in (1, 11, 12, 2, 3, 13, 14, 4, 5, 15, 16, 6, 7, 8, 9, 10):
    v = rel[(code == c) & valid]
    if v.size: print("  %-36s %6.2f %6.2f %6.2f   (%d)" % (names[c], v.min(), np.median(v), v.max(), v.size))
for mode in (1, 0, 1, 0):
    net.set_option("act_kernel", mode)
    for _ in range(200): buf.add(rng.randint(0, 256, (84, 84), dtype=np.uint8)); net.predict_state(buf)
    ts = []
    for _ in range(2000):
        buf.add(rng.randint(0, 256, (84, 84), dtype=np.uint8))
        t = time.perf_counter(); net.predict_state(buf); ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e6
    print("act_kernel=%d: add; predict_state median %.1f us  p10 %.1f  p90 %.1f" % (mode, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))

for mode in (1, 0):
    net.set_option("act_kernel", mode)
    ts = []
    for _ in range(1000):
        buf.add(rng.randint(0, 256, (84, 84), dtype=np.uint8)); net.sync()          # the frame's H2D copy has completed
        t = time.perf_counter(); net.predict_state(buf); ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e6
    print("act_kernel=%d: add; SYNC; predict_state median %.1f us  p10 %.1f  p90 %.1f" % (mode, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
