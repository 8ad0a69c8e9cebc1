"""Diagnostic (not product): bench.py's `q_mae_vs_cpu_ref` leg step by step, on the state a `--steps K --warmup W` run leaves behind.
For each of the 10 comparison steps: free-running error (oracle and library both continue from their own state) and teacher-forced
error (oracle re-loaded with the library's theta / theta- / s before the step), plus the number of ReLU gates (a1..a4, online net)
that differ between the two implementations on the step's minibatch.   K=20 W=5 [TPS2=14] python tools/exp/qmae_diag.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from simple_dqn_amd import _lib
from util import make_args
from bench import fill_ring, oracle_view
from oracle.dqn_numpy import OracleDQN
from oracle.replay_numpy import MT19937
B, A, seed = 32, 4, 123
K, W = int(os.environ.get("K", 20)), int(os.environ.get("W", 5))
args = make_args(batch_size=B, random_seed=seed + 1)
mem = sd.ReplayMemory(int(os.environ.get("RING", 1000000)), args); fill_ring(mem, seed, A)
net = sd.DeepQNetwork(A, args); net.update_target_network()
if os.environ.get("TPS2"): net.set_option("tps:2", int(os.environ["TPS2"]))
if os.environ.get("R1"):                     # round 1's summation orders / dispatch order
    net.set_option("tps:2", 14); net.set_option("nw:2", 9); net.set_option("bwd_order", 1)
mt = (C.c_uint32 * 625)(); _lib.check(sd.load().sdqn_mt_seed(mt, seed + 2))
net.train_from_memory(mem, max(W - 2, 3) + 2 + K, mt_state=mt, want_cost=False); net.sync()


def load(o):
    o.W = [w.copy() for w in net.get_weights(0)]; o.Wt = [w.copy() for w in net.get_weights(1)]; o.S = [w.copy() for w in net.get_weights(2)]


free = OracleDQN(A, batch_size=B, weights=net.get_weights(0)); load(free)
tf = OracleDQN(A, batch_size=B, weights=net.get_weights(0))
omem = oracle_view(mem, B)
rng = MT19937(); rng.setstate(tuple(mt[:]))
hold_rng = MT19937(); hold_rng.setstate(tuple(mt[:]))
for _ in range(40): hold = omem.getMinibatch(hold_rng)[0].copy()
for s in range(int(os.environ.get("STEPS", 10))):
    mb = [x.copy() for x in omem.getMinibatch(rng)]
    load(tf)
    net.train_from_memory(mem, 1, mt_state=mt, want_cost=False); net.sync()
    free.train(mb); tf.train(mb)
    assert tuple(mt[:]) == rng.getstate()
    q = net.predict(hold)
    ef, et = np.abs(q - free.predict(hold)), np.abs(q - tf.predict(hold))
    dW = [float(np.abs(a - b).max()) for a, b in zip(net.get_weights(0), tf.W)]
    print("step %2d  free-running mae %.2e max %.2e | teacher-forced mae %.2e max %.2e | max |dW| per layer after the forced step %s"
          % (s, ef.mean(), ef.max(), et.mean(), et.max(), ["%.1e" % x for x in dW]), flush=True)
