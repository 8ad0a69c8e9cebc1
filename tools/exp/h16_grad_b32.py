import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights, OracleDQN
A, B = 4, 32
for seed in (411, 511, 611):
    mb = random_minibatch(B, A, seed + 1, reward_range=(-2, 3))
    for name, opts in (("chain+c1", []), ("chain", [("bt:0", 6)]), ("old", [("bt:1", 6), ("bt:2", 6)])):
        for ex in (True, False):
            net = sd.DeepQNetwork(A, make_args(batch_size=B, datatype="float16"))
            ws, wt = xavier_weights(A, seed), xavier_weights(A, seed + 1)
            net.set_weights(wt, 1); net.set_weights(ws, 0); net.set_option("keep_gradients", 1)
            for k, v in opts: net.set_option(k, v)
            o = OracleDQN(A, batch_size=B, weights=ws, half_activations=True, exact_conv1_input=ex); o.Wt = [w.copy() for w in wt]
            g = o.gradients(mb)[0]
            net.train(mb)
            r = []
            for i in range(5):
                gg = net.get_layer(i, which=3)
                r.append("%.1e/%.1e" % (np.abs(gg - g[i]).max() / max(1e-6, np.abs(g[i]).max()), np.linalg.norm(gg - g[i]) / np.linalg.norm(g[i])))
            print(seed, "%-9s oracle exact=%-5s max-rel/fro per layer:" % (name, ex), " ".join(r))
