import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args, random_minibatch
B, A = 256, 3
mb = random_minibatch(B, A, 3)
def run(opts):
    net = sd.DeepQNetwork(A, make_args(batch_size=B))
    net.set_option("fused_launches", 0)
    for k, v in opts: net.set_option(k, v)
    for _ in range(3): net.train(mb)
    net.profile(True, -1); net.profile_reset()
    for _ in range(10): net.train(mb)
    r = {p["name"].split("(")[0]: round(p["total_ms"] / p["launches"] * 1e3, 1) for p in net.profile_read() if p["launches"]}
    net.profile(False)
    return r
print("default (staged where R1 chose)", run([]))
print("direct nw=8 for conv2_fwd conv3_fwd fc4_fwd fc4_dgrad conv3_dgrad conv2_dgrad", run([("nw:1", 8), ("nw:2", 8), ("nw:3", 8), ("nw:5", 4), ("nw:7", 8), ("nw:9", 8)]))
