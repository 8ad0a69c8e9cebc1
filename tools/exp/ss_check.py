"""Round 6: the sample-stationary forward convolutions (csrc/conv_ss.h) against the block-tile routine they replace (bt:<id> = 6):
intermediate activations, Q-values, per-launch time, run-to-run bit stability.  usage: python tools/exp/ss_check.py [B ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights


def net_of(A, B, opts=()):
    n = sd.DeepQNetwork(A, make_args(batch_size=B))
    n.set_weights(xavier_weights(A, 8), 1)
    n.set_weights(xavier_weights(A, 7), 0)
    for k, v in opts:
        n.set_option(k, v)
    return n


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


def main():
    Bs = [int(x) for x in sys.argv[1:]] or [256, 160, 136, 128]
    for B in Bs:
        A = 3
        mb = random_minibatch(B, A, 40 + B, reward_range=(-2, 3))
        new = net_of(A, B, [("keep_gradients", 1), ("bt:1", 7), ("bt:2", 7)])          # both layers: ONE chained launch
        new2 = net_of(A, B, [("keep_gradients", 1), ("bt:1", 8), ("bt:2", 8)])         # the same routine, two launches
        old = net_of(A, B, [("keep_gradients", 1), ("bt:1", 6), ("bt:2", 6)])
        for n in (new, new2, old):
            n.train(mb)
        for name, cnt in dict(a1=2 * B * 400 * 32, a2=2 * B * 81 * 64, a3=2 * B * 49 * 64, a4=2 * B * 512).items():
            x, x2, y = new.debug_read(name, cnt), new2.debug_read(name, cnt), old.debug_read(name, cnt)
            bad = np.flatnonzero(np.abs(x - y) > 1e-4 * max(1e-6, np.abs(y).max()))
            print("B=%d %s rel err vs block-tile %.3e  chained == two launches %s  nbad %d first %s" % (B, name, rel(x, y), np.array_equal(x, x2), bad.size, bad[:6]))
        print("B=%d q max abs diff %.3e" % (B, np.abs(new.last_q()[0] - old.last_q()[0]).max()))
        for tag, n in (("ss-chain", new), ("ss-2launch", new2), ("bt", old), ("ss-chain", new), ("ss-2launch", new2), ("bt", old)):
            for _ in range(20):
                n.train(mb)
            n.profile(True, -1); n.profile_reset()
            for _ in range(40):
                n.train(mb)
            prof = {p["name"]: p["total_ms"] / p["launches"] * 1e3 for p in n.profile_read() if p["launches"] >= 40}
            n.profile(False)
            print("B=%d %s: " % (B, tag) + "  ".join("%s %.2f" % (k.split("(")[0], v) for k, v in prof.items()))


if __name__ == "__main__":
    main()
