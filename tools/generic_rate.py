"""Step rate of the generic im2col + GEMM path (csrc/generic_net.hip): float64 at the headline geometry, float32 at another geometry,
fused loop (native sampler -> device gather -> step) from a device-resident ring."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args
for dtype, hist, H, W, B in (("float64", 4, 84, 84, 32), ("float32", 4, 84, 84, 32), ("float32", 4, 96, 96, 32), ("float64", 4, 84, 84, 256)):
    if (dtype, hist, H, W) == ("float32", 4, 84, 84):
        continue                                     # (that is the tuned path)
    args = make_args(batch_size=B, history_length=hist, screen_height=H, screen_width=W, datatype=dtype)
    mem = sd.ReplayMemory(20000, args)
    rng = np.random.RandomState(1)
    mem.screens[:] = rng.randint(0, 256, size=mem.screens.shape, dtype=np.uint8); mem.actions[:] = rng.randint(0, 4, size=mem.size)
    mem.rewards[:] = rng.randint(-1, 2, size=mem.size); mem.terminals[:] = rng.rand(mem.size) < 0.005
    mem.count, mem.current = mem.size, mem.size // 3
    net = sd.DeepQNetwork(4, args); net.update_target_network()
    mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
    net.train_from_memory(mem, 20, mt_state=mt, want_cost=False); net.sync()
    N = 200 if B <= 32 else 40
    t = time.perf_counter(); net.train_from_memory(mem, N, mt_state=mt, want_cost=False); net.sync(); dt = time.perf_counter() - t
    flops = 2 * 34103296 * B * (H * W) / (84 * 84)          # (scaled roughly with the screen area)
    print("%s %dx%dx%d B=%d: %.0f steps/s, %.2f ms/step, ~%.1f TFLOP/s" % (dtype, hist, H, W, B, N / dt, dt / N * 1e3, flops * N / dt / 1e12))
