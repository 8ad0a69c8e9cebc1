"""Experiment (not product): the step rate (B = 32; B=256 A=3 in the environment: BASELINE configs[2]) against the size of the replay ring — how much of the step waits for ring frames that
come from HBM (1 M frames = 7 GB: every sampled frame is cold) rather than from the memory-side cache (20 k frames = 141 MB) or L2
(500 frames = 3.5 MB).  An upper bound for what prefetching the next minibatch's frames could buy.   python tools/exp/ring_size_rate.py"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = int(os.environ.get("B", 32)), int(os.environ.get("A", 4))
N = 3000 if B <= 64 else 600
args = make_args(batch_size=B)
SIZES = [int(x) for x in os.environ.get("SIZES", "1000000,500,20000,1000000,500,20000,2000,100000").split(",")]
for size in SIZES:
    mem = sd.ReplayMemory(size, args); fill_ring(mem, 1, A)
    net = sd.DeepQNetwork(A, args); net.update_target_network()
    mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
    net.train_from_memory(mem, N // 5, mt_state=mt, want_cost=False); net.sync()
    r = []
    for _ in range(4):
        t = time.perf_counter(); net.train_from_memory(mem, N, mt_state=mt, want_cost=False); net.sync()
        r.append(N / (time.perf_counter() - t))
    print("ring %8d frames (%7.1f MB): %s max %d steps/s = %.2f us/step" % (size, size * 7056 / 1e6, [round(x) for x in r], max(r), 1e6 / max(r)), flush=True)
    del net, mem
