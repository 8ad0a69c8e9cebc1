import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights
for B in (256, 129, 160):
    A = 3
    mb = random_minibatch(B, A, 77 + B, reward_range=(-2, 3))
    nets = {}
    for m in (0, 1, 2):
        n = sd.DeepQNetwork(A, make_args(batch_size=B, datatype="float16"))
        n.set_weights(xavier_weights(A, 8), 1); n.set_weights(xavier_weights(A, 7), 0)
        n.set_option("keep_gradients", 1); n.set_option("c1w_in_wgrads", m)
        n.train(mb); n.train(mb)
        nets[m] = n
    for m in (1, 2):
        print("B=%d c1w_in_wgrads=%d: gradients identical to 0: %s" % (B, m, [bool(np.array_equal(nets[m].get_layer(i, 3), nets[0].get_layer(i, 3))) for i in range(5)]), "max|g0|", float(np.abs(nets[m].get_layer(0, 3)).max()))
    for rep in range(2):
        for m in (0, 1, 2):
            n = nets[m]
            for _ in range(20): n.train(mb)
            n.profile(True, -1); n.profile_reset()
            for _ in range(40): n.train(mb)
            prof = {p["name"].split("(")[0]: p["total_ms"] / p["launches"] * 1e3 for p in n.profile_read() if p["launches"] >= 40}
            n.profile(False)
            print("B=%d mode %d: " % (B, m) + "  ".join("%s %.2f" % kv for kv in prof.items() if kv[0] in ("wgrads", "bwd1", "update")))
