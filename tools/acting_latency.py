"""Experiment (not product): per-env-step latency of the three acting paths."""
import os, time, numpy as np, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import simple_dqn_amd as sd
from util import make_args
args = make_args(batch_size=32)
net = sd.DeepQNetwork(4, args)
ref, dev = sd.StateBuffer(args), sd.DeviceStateBuffer(args)
rng = np.random.RandomState(0)
frames = rng.randint(0, 256, size=(64, 84, 84), dtype=np.uint8)
for name, fn in (("predict(padded minibatch)", lambda i: (ref.add(frames[i % 64]), net.predict(ref.getStateMinibatch())[0])),
                 ("predict_one(host state)", lambda i: (ref.add(frames[i % 64]), net.predict_one(ref.getState()))),
                 ("predict_state(device buffer)", lambda i: (dev.add(frames[i % 64]), net.predict_state(dev)))):
    for i in range(50): fn(i)
    t = time.perf_counter()
    for i in range(2000): fn(i)
    print("%-32s %.1f us per env step (add + Q-values)" % (name, (time.perf_counter() - t) / 2000 * 1e6))
