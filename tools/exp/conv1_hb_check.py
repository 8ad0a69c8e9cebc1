"""Round 6: float16 conv1 forward at B >= 128 — the exact-byte form (conv1_hb_kernel, bt:0 = 0 write-through / 2 plain stores) against
the first form (conv1_h_kernel, half(b / 255) operands, bt:0 = 1): Q-values against each other and the half oracle, per-launch time.
usage: python tools/exp/conv1_hb_check.py [B ...]"""
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
    for k, v in opts:
        n.set_option(k, v)
    return n


def main():
    Bs = [int(x) for x in sys.argv[1:]] or [256, 160, 128]
    for B in Bs:
        A = 3
        mb = random_minibatch(B, A, 40 + B, reward_range=(-2, 3))
        nets = [("exact-wt", net_of(A, B)), ("exact-plain", net_of(A, B, [("bt:0", 2)])), ("first", net_of(A, B, [("bt:0", 1), ("bt:18", 1)]))]
        o = OracleDQN(A, batch_size=B, weights=xavier_weights(A, 7), half_activations=True)
        o32 = OracleDQN(A, batch_size=B, weights=xavier_weights(A, 7))
        qo, q32 = o.predict(mb[0]), o32.predict(mb[0])
        qs = {k: n.predict(mb[0]).copy() for k, n in nets}
        for k in qs:
            print("B=%d %-11s: max|q - half oracle| %.3e   max|q - fp32 oracle| %.3e   max|q - first form| %.3e   bit-stable %s" % (
                B, k, np.abs(qs[k] - qo).max(), np.abs(qs[k] - q32).max(), np.abs(qs[k] - qs["first"]).max(),
                np.array_equal(qs[k], dict(nets)[k].predict(mb[0]))))
        print("B=%d half oracle vs fp32 oracle %.3e" % (B, np.abs(qo - q32).max()))
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
