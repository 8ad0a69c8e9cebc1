"""Experiment (not product): where an acting step's time goes.  python tools/exp/acting_breakdown.py"""
import os, time, numpy as np, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args
args = make_args(batch_size=32)
net = sd.DeepQNetwork(4, args)
dev = sd.DeviceStateBuffer(args)
rng = np.random.RandomState(0)
frames = rng.randint(0, 256, size=(64, 84, 84), dtype=np.uint8)
def rate(fn, n=3000):
    for i in range(100): fn(i)
    net.sync(); t = time.perf_counter()
    for i in range(n): fn(i)
    net.sync(); return (time.perf_counter() - t) / n * 1e6
print("dev.add only                 %.1f us" % rate(lambda i: dev.add(frames[i % 64])))
print("predict_state only           %.1f us" % rate(lambda i: net.predict_state(dev)))
print("add + predict_state          %.1f us" % rate(lambda i: (dev.add(frames[i % 64]), net.predict_state(dev))))
if hasattr(net, "act_step"):
    print("act_step (add only)          %.1f us" % rate(lambda i: net.act_step(dev, None, frames[i % 64])))
    print("act_step(speculate) + predict_state %.1f us" % rate(lambda i: (net.act_step(dev, None, frames[i % 64], speculate=True), net.predict_state(dev))))
    def spec_with_work(i):
        net.act_step(dev, None, frames[i % 64], speculate=True)
        t = time.perf_counter()
        while time.perf_counter() - t < 20e-6: pass                      # ~20 us of "environment" between the transition and the next action
        net.predict_state(dev)
    print("... with 20 us of host work in between: %.1f us (i.e. %.1f us beyond the host work)" % (rate(spec_with_work), rate(spec_with_work) - 20))
net.profile(True, -1); net.profile_reset()
for i in range(200): net.predict_state(dev)
print({p["name"]: round(p["total_ms"] / p["launches"] * 1e3, 2) for p in net.profile_read() if p["launches"]})
net.profile(False)
