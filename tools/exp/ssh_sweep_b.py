"""Round 6 safety net: the float16 chains (default) against the launches they replace (bt:0/1/2/7/9 = 6) over odd batch sizes, one train step:
Q-values, cost, all five gradients.  B >= 128: bit-identical except where conv1's input semantics differ (first-form conv1 when forced);
below: tolerance.  usage: python tools/exp/ssh_sweep_b.py [B ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights


def net_of(A, B, opts=()):
    n = sd.DeepQNetwork(A, make_args(batch_size=B, datatype="float16"))
    n.set_weights(xavier_weights(A, 8), 1); n.set_weights(xavier_weights(A, 7), 0)
    n.set_option("keep_gradients", 1)
    for k, v in opts:
        n.set_option(k, v)
    return n


def rf(a, b):
    return float(np.linalg.norm(a - b) / max(1e-12, np.linalg.norm(b)))


bad = 0
for B in [int(x) for x in sys.argv[1:]] or [2, 7, 31, 33, 47, 49, 97, 127, 130, 161, 200, 257, 300, 511, 512, 600]:
    A = 3 + B % 4
    mb = random_minibatch(B, A, 900 + B, reward_range=(-2, 3))
    new = net_of(A, B)
    old = net_of(A, B, [("bt:1", 6), ("bt:2", 6), ("bt:7", 6), ("bt:9", 6), ("bt:0", 2 if B >= 48 else 6)])    # conv1: the exact-byte kernel on its own where it exists
    q1, q2 = new.predict(mb[0]).copy(), old.predict(mb[0]).copy()
    new.train(mb); old.train(mb)
    g = [rf(new.get_layer(i, 3), old.get_layer(i, 3)) for i in range(5)]
    same = np.array_equal(q1, q2) and all(np.array_equal(new.get_layer(i, 3), old.get_layer(i, 3)) for i in range(5))
    ok = np.isfinite(q1).all() and np.abs(q1 - q2).max() < 1e-3 and max(g) < 5e-2
    bad += not ok
    print("B=%3d A=%d: max|dq| %.2e  grads rel Fro %s  bit-identical %s  %s" % (B, A, np.abs(q1 - q2).max(), " ".join("%.1e" % x for x in g), same, "ok" if ok else "MISMATCH"))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
