"""Bring-up check (not product): fp16-mode weight gradients, packed-fp16 MFMA routine (h16_wgrad_mfma=1) vs the round-1
fp32-MFMA routine with half operands (=0) vs the half oracle, one step from identical weights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import OracleDQN, xavier_weights
for A, B in ((4, 32), (6, 32), (3, 256)):
    ws, wt = xavier_weights(A, 11), xavier_weights(A, 12)
    mb = random_minibatch(B, A, 13, reward_range=(-2, 3))
    o = OracleDQN(A, batch_size=B, weights=ws, half_activations=True); o.Wt = [w.copy() for w in wt]
    g, cost, _, preq = o.gradients(mb)
    res = {}
    for mode in (0, 1):
        net = sd.DeepQNetwork(A, make_args(batch_size=B, datatype="float16"))
        net.set_weights(wt, 1); net.set_weights(ws, 0)
        net.set_option("keep_gradients", 1); net.set_option("h16_wgrad_mfma", mode)
        net.train(mb)
        res[mode] = [net.get_layer(i, 3) for i in range(5)]
    for i in range(5):
        sc = max(1e-6, np.abs(g[i]).max())
        print("A=%d B=%d layer %d: |new-old|/max %.2e   |old-oracle|/max %.2e   |new-oracle|/max %.2e   frac of elements with |new-old| > 1e-4 max: %.4f"
              % (A, B, i, np.abs(res[1][i] - res[0][i]).max() / sc, np.abs(res[0][i] - g[i]).max() / sc, np.abs(res[1][i] - g[i]).max() / sc,
                 float((np.abs(res[1][i] - res[0][i]) > 1e-4 * sc).mean())))
