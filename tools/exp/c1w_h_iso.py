import sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights, OracleDQN
def rf(a, b): return float(np.linalg.norm(a - b) / max(1e-12, np.linalg.norm(b)))
for B in (128, 256):
    A = 3
    mb = random_minibatch(B, A, 750 + B, reward_range=(-2, 3))
    g = {}
    for f in (0, 1):
        for w in (0, 1):
            n = sd.DeepQNetwork(A, make_args(batch_size=B, datatype="float16"))
            n.set_weights(xavier_weights(A, 8), 1); n.set_weights(xavier_weights(A, 7), 0)
            n.set_option("keep_gradients", 1); n.set_option("bt:0", f); n.set_option("bt:18", w)
            n.train(mb)
            g[(f, w)] = n.get_layer(0, which=3).copy()
    print("B=%d fwd exact: wgrad exact vs first %.3e | fwd first: wgrad exact vs first %.3e | wgrad first: fwd exact vs first %.3e | wgrad exact: fwd exact vs first %.3e" % (
        B, rf(g[(0, 0)], g[(0, 1)]), rf(g[(1, 0)], g[(1, 1)]), rf(g[(0, 1)], g[(1, 1)]), rf(g[(0, 0)], g[(1, 0)])))
    for ex in (True, False):
        o = OracleDQN(A, batch_size=B, weights=xavier_weights(A, 7), half_activations=True, exact_conv1_input=ex)
        o.Wt = [x.copy() for x in xavier_weights(A, 8)]
        go = o.gradients(mb)[0][0]
        print("  oracle exact=%s: " % ex + "  ".join("%s %.3e" % (k, rf(v, go)) for k, v in g.items()), " |g| %.3e" % np.linalg.norm(go))
