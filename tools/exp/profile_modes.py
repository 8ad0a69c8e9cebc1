"""Per-kernel live timing, two ways (option "profile_mode"): 1 = the launch records its own dispatch-packet timestamps
(hipExtLaunchKernel start / stop events), 0 = hipEventRecord markers around the launch.  Prints both per kernel, and what
timing every N-th launch of the dominant kernel costs the step rate in each mode."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = int(os.environ.get("B", 32)), 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
def run(n):
    net.train_from_memory(mem, n, mt_state=mt, want_cost=False); net.sync()
run(600)
res = {}
for mode in (1, 0, 1, 0):
    net.set_option("profile_mode", mode); net.set_option("profile_every", 1)
    net.profile(True, -1); net.profile_reset(); run(400)
    res[mode] = {p["name"]: p["total_ms"] / p["launches"] * 1e3 for p in net.profile_read() if p["launches"] >= 400}
    net.profile(False)
print("%-50s %10s %10s" % ("kernel (us per launch, every launch timed)", "packet ts", "markers"))
for k in res[1]:
    print("%-50s %10.2f %10.2f" % (k, res[1][k], res[0].get(k, float("nan"))))
print("%-50s %10.2f %10.2f" % ("sum", sum(res[1].values()), sum(res[0].values())))
dom = 16
N = 4000
def rate():
    t = time.perf_counter(); run(N); return N / (time.perf_counter() - t)
print("step rate, nothing timed: %.0f %.0f" % (rate(), rate()))
for mode in (1, 0):
    for every in (64, 16, 4, 1):
        net.set_option("profile_mode", mode); net.set_option("profile_every", every)
        net.profile(True, dom); net.profile_reset()
        r = [rate(), rate()]
        p = [q for q in net.profile_read() if q["id"] == dom][0]
        net.profile(False)
        print("mode %d every %2d: %.0f %.0f steps/s; bwd3 %.2f us over %d launches" % (mode, every, r[0], r[1], p["total_ms"] / max(p["launches"], 1) * 1e3, p["launches"]))
