"""Round 6: does the float16 forward chain (conv_ssh.h) pay BELOW the throughput regime?  B = 32 / 64: default launches against the forced
chain (bt:1 = bt:2 = 7; 2 B / 1 workgroups on 256 CUs): Q agreement, per-launch time, step rate.  usage: python tools/exp/ssh_b32.py [B ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights


def net_of(A, B, opts=()):
    n = sd.DeepQNetwork(A, make_args(batch_size=B, datatype="float16"))
    n.set_weights(xavier_weights(A, 8), 1)
    n.set_weights(xavier_weights(A, 7), 0)
    for k, v in opts:
        n.set_option(k, v)
    return n


def main():
    for B in [int(x) for x in sys.argv[1:]] or [32, 64]:
        A = 4
        mb = random_minibatch(B, A, 40 + B, reward_range=(-2, 3))
        nets = [("default", net_of(A, B)), ("chain", net_of(A, B, [("bt:1", 7), ("bt:2", 7)]))]
        qs = {k: n.predict(mb[0]).copy() for k, n in nets}
        print("B=%d max|q chain - q default| %.3e (|q| max %.3f)" % (B, np.abs(qs["chain"] - qs["default"]).max(), np.abs(qs["default"]).max()))
        for rep in range(3):
            for tag, n in nets:
                for _ in range(50):
                    n.train(mb)
                n.sync(); t0 = time.perf_counter()
                for _ in range(1000):
                    n.train(mb)
                n.sync(); dt = time.perf_counter() - t0
                n.profile(True, -1); n.profile_reset()
                for _ in range(40):
                    n.train(mb)
                prof = {p["name"]: p["total_ms"] / p["launches"] * 1e3 for p in n.profile_read() if p["launches"] >= 40}
                n.profile(False)
                print("B=%d %-8s %7.0f train(tuple) steps/s : " % (B, tag, 1000 / dt) + "  ".join("%s %.2f" % (k.split("(")[0], v) for k, v in prof.items()))


if __name__ == "__main__":
    main()
