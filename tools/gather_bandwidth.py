"""Experiment (not product): standalone replay gather (sdqn_replay_gather, u8 out) bandwidth vs batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
for B in (32, 256, 1024, 4096):
    args = make_args(batch_size=B)
    mem = sd.ReplayMemory(200000, args); fill_ring(mem, 1, 4)
    idx = np.random.RandomState(0).randint(4, 200000, B).astype(np.int64)
    ms = mem.bench_gather(idx, iters=200)
    by = B * 13 * 7056
    print("B %5d: %8.2f us per gather, algorithmic %7.2f MB -> %6.2f TB/s (%4.1f %% of 8 TB/s spec, %4.1f %% of the measured 4.64 TB/s triad)"
          % (B, ms * 1e3, by / 1e6, by / ms / 1e9, by / ms / 1e9 / 8 * 100, by / ms / 1e9 / 4.641 * 100), flush=True)
    del mem
