import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights
A, B = 3, 256
mb = random_minibatch(B, A, 5)
for opts in ([("fused_launches", 0)], []):
    n = sd.DeepQNetwork(A, make_args(batch_size=B))
    for k, v in opts: n.set_option(k, v)
    for _ in range(20): n.train(mb)
    n.profile(True, -1); n.profile_reset()
    for _ in range(40): n.train(mb)
    prof = {p["name"]: p["total_ms"] / p["launches"] * 1e3 for p in n.profile_read() if p["launches"] >= 40}
    n.profile(False)
    print(opts, "  ".join("%s %.2f" % (k.split("(")[0], v) for k, v in prof.items()))
