import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, simple_dqn_amd as sd
from util import make_args, random_minibatch
B, A = 256, 3
mb = random_minibatch(B, A, 3)
net = sd.DeepQNetwork(A, make_args(batch_size=B)); net.update_target_network()
for o in sys.argv[1:]:
    k, v = o.split("="); net.set_option(k, int(v))
for _ in range(5): net.train(mb)
net.profile(True, 18); net.profile_reset()
for _ in range(40): net.train(mb)
print(os.environ.get("SDQN_C1W_DBG", "0"), sys.argv[1:], [round(p["total_ms"] / p["launches"] * 1e3, 1) for p in net.profile_read() if p["launches"]])
