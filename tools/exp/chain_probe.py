"""Experiment: the training forward conv chain conv1 -> conv2 -> conv3 as ONE XCC-local launch (sdqn_act.hip:
chain_probe_kernel, experiments build), ns (state, net) pairs per XCC — 8 = batch 32 with both nets — against the three forward launches
of the product step.   SDQN_LIB_VARIANT=experiments python tools/exp/chain_probe.py
Historical (round 4): the probe and the experiments build left the library in round 5 — runs on the tree that
tools/exp/experiments_r04.patch re-creates (on commit 4b59432); the result is in tools/exp/README.md."""
import os, sys, ctypes as C
os.environ.setdefault("SDQN_LIB_VARIANT", "experiments")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from simple_dqn_amd import _lib
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights
GHZ = float(os.environ.get("GHZ", 2.4))
lib = sd.load()
net = sd.DeepQNetwork(4, make_args(batch_size=32)); net.set_weights(xavier_weights(4, 1), 0); net.update_target_network()
# the product step's three forward launches (dispatch-packet durations, every launch bracketed)
mb = random_minibatch(32, 4, 3)
for _ in range(20): net.train(mb)
net.set_option("profile_every", 1); net.profile(True, -1); net.profile_reset()
for _ in range(200): net.train(mb)
us = {p["id"]: p["total_ms"] / p["launches"] * 1e3 for p in net.profile_read() if p["launches"]}
net.profile(False)
print("product step, B = 32, both nets: conv1_fwd %.2f + conv2_fwd %.2f + conv3_fwd %.2f = %.2f us (three launches)" % (us[0], us[1], us[2], us[0] + us[1] + us[2]))
names = {1: "conv1 item start", 2: "conv1 item signalled", 3: "conv2 item start (its pair's conv1 complete)", 4: "conv2 item signalled",
         5: "conv3 item start (its pair's conv2 complete)", 6: "conv3 item signalled", 11: "conv1 operands landed", 12: "conv1 MFMA done",
         13: "conv2 operands landed", 14: "conv2 MFMA done", 15: "conv3 operands landed", 16: "conv3 MFMA done"}
for ns in (1, 2, 4, 8):          # (-8: static tickets — only valid where workgroup b lands on XCC b % 8; it did not on the MI355X boxes of this round: the launch aborts)
    for grid in (256, 512, 1024):
        t = C.c_float(); st = np.zeros((grid, 80), np.uint64)
        _lib.check(lib.sdqn_exp_chain_probe(net._h, ns, grid, 30, C.byref(t), st.ctypes.data_as(C.POINTER(C.c_uint64))))
        kinds = st[:, 0:78:2].astype(np.int64); clk = st[:, 1:79:2].astype(np.int64)
        code = kinds >> 16
        rel = (clk - clk[:, :1]) / (GHZ * 1e3)
        valid = np.arange(39)[None, :] < (kinds != 0).sum(1)[:, None] + 1
        end = rel[(code == 6) & valid]
        print("chain probe%s: %d pair(s) per XCC (%2d conv chains), %4d workgroups: %6.2f us per launch; last conv3 item signalled %.2f us after its workgroup's start"
              % (" (STATIC tickets)" if ns < 0 else "", abs(ns), 8 * abs(ns), grid, t.value, end.max() if end.size else float("nan")))
        if abs(ns) == 8 and grid == int(os.environ.get("DETAIL_GRID", 1024)):
            print("  time after the workgroup's own start, us: min / median / max over the launch's items (stamps carry a forced s_waitcnt: the stamped launch is a little slower)")
            for c in (1, 11, 12, 2, 3, 13, 14, 4, 5, 15, 16, 6):
                v = rel[(code == c) & valid]
                if v.size: print("    %-46s %6.2f %6.2f %6.2f   (%d)" % (names[c], v.min(), np.median(v), v.max(), v.size))
