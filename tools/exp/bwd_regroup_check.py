"""Round 6: fp32, B >= 128 — the backward launches regrouped (option bwd_regroup: bwd3 = conv3_dgrad || fc4_wgrad, bwd2 = conv3_wgrad || conv2_dgrad,
bwd1 = conv2_wgrad || conv1_wgrad) against the built-in grouping: gradients (same tiles: same bits), per-launch time."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights
for B in [int(x) for x in sys.argv[1:]] or [256, 129]:
    A = 3
    mb = random_minibatch(B, A, 70 + B, reward_range=(-2, 3))
    nets = {}
    for m in (0, 1):
        n = sd.DeepQNetwork(A, make_args(batch_size=B))
        n.set_weights(xavier_weights(A, 8), 1); n.set_weights(xavier_weights(A, 7), 0)
        n.set_option("bwd_regroup", m)
        n.train(mb); n.train(mb)
        nets[m] = n
    print("B=%d regrouped == built-in (weights after two steps): %s" % (B, [bool(np.array_equal(a, b)) for a, b in zip(nets[1].get_weights(0), nets[0].get_weights(0))]))
    for rep in range(2):
        for m in (0, 1):
            n = nets[m]
            for _ in range(20): n.train(mb)
            n.profile(True, -1); n.profile_reset()
            for _ in range(40): n.train(mb)
            prof = {p["name"].split("(")[0]: p["total_ms"] / p["launches"] * 1e3 for p in n.profile_read() if p["launches"] >= 40}
            n.profile(False)
            print("B=%d regroup %d: " % (B, m) + "  ".join("%s %.2f" % kv for kv in prof.items() if kv[0].startswith("bwd") or kv[0] in ("update", "fc4_dgrad")), " sum bwd %.2f" % sum(v for k, v in prof.items() if k.startswith("bwd")))
