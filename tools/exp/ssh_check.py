"""Round 6: float16 conv2 -> conv3 forward as one sample-stationary launch (csrc/conv_ssh.h; bt:1 = bt:2 = 0 / 7 / 8) against the packed-fp16
block-tile routines (bt:1 = bt:2 = 6): Q-values, gradients, per-launch time.  usage: python tools/exp/ssh_check.py [B ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights, OracleDQN


def net_of(A, B, opts=()):
    n = sd.DeepQNetwork(A, make_args(batch_size=B, datatype="float16"))
    n.set_weights(xavier_weights(A, 8), 1)
    n.set_weights(xavier_weights(A, 7), 0)
    n.set_option("keep_gradients", 1)
    for k, v in opts:
        n.set_option(k, v)
    return n


def rf(a, b):
    return float(np.linalg.norm(a - b) / max(1e-12, np.linalg.norm(b)))


def main():
    Bs = [int(x) for x in sys.argv[1:]] or [256, 129, 128]
    for B in Bs:
        A = 3
        mb = random_minibatch(B, A, 40 + B, reward_range=(-2, 3))
        ids = (1, 2, 7, 9)               # conv2_fwd, conv3_fwd, conv3_dgrad, conv2_dgrad
        nets = [("chain", net_of(A, B, [("bt:%d" % i, 7) for i in ids])), ("chain-plain", net_of(A, B, [("bt:%d" % i, 8) for i in ids])),
                ("bt", net_of(A, B, [("bt:%d" % i, 6) for i in ids])), ("fwd-only", net_of(A, B, [("bt:7", 6), ("bt:9", 6)])), ("default", net_of(A, B))]
        o = OracleDQN(A, batch_size=B, weights=xavier_weights(A, 7), half_activations=True)
        o.Wt = [w.copy() for w in xavier_weights(A, 8)]
        qo = o.predict(mb[0])
        go = o.gradients(mb)[0]
        qs = {k: n.predict(mb[0]).copy() for k, n in nets}
        gs = {}
        for k, n in nets:
            n.train(mb)
            gs[k] = [n.get_layer(i, which=3).copy() for i in range(5)]
        for k, n in nets:
            print("B=%d %-11s: max|q - oracle| %.3e  max|q - bt| %.3e  stable %s  grads vs bt (rel Fro) %s  vs oracle %s" % (
                B, k, np.abs(qs[k] - qo).max(), np.abs(qs[k] - qs["bt"]).max(), np.array_equal(qs[k], n.predict(mb[0])) if False else "-",
                " ".join("%.1e" % rf(gs[k][i], gs["bt"][i]) for i in range(5)), " ".join("%.1e" % rf(gs[k][i], go[i]) for i in range(5))))
        for rep in range(2):
            for tag, n in nets:
                for _ in range(20):
                    n.train(mb)
                n.profile(True, -1); n.profile_reset()
                for _ in range(40):
                    n.train(mb)
                prof = {p["name"]: p["total_ms"] / p["launches"] * 1e3 for p in n.profile_read() if p["launches"] >= 40}
                n.profile(False)
                print("B=%d %-11s: " % (B, tag) + "  ".join("%s %.2f" % (k.split("(")[0], v) for k, v in prof.items()))


if __name__ == "__main__":
    main()
