"""Experiment (not product): where the host time of the reference-style loop net.train(mem.getMinibatch()) goes (cProfile, B = 32)."""
import cProfile, os, pstats, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = 32, 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
random.seed(1)
for _ in range(200): net.train(mem.getMinibatch())
net.sync()
def loop(n):
    for _ in range(n): net.train(mem.getMinibatch())
    net.sync()
t = time.perf_counter(); loop(3000); print("plain: %.1f us per iteration" % ((time.perf_counter() - t) / 3000 * 1e6))
pr = cProfile.Profile(); pr.enable(); loop(3000); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
