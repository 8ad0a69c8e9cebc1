"""[historical, rounds 2-4: the float32 register-blocked routine and its option "rb:<id>" left the library in round 5 — runs on the
tree of tools/exp/experiments_r04.patch]  Bring-up helper (not product): run one configuration of the B=256 step N times (for rocprofv3 --pmc / --kernel-trace).
   python tools/rb_probe.py "rb:1=2,rb:2=4" [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args, random_minibatch
B, A = 256, 3
opts = [kv.split("=") for kv in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1] else []
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
net = sd.DeepQNetwork(A, make_args(batch_size=B))
net.set_option("fused_launches", 0)
for k, v in opts:
    net.set_option(k, int(v))
mb = random_minibatch(B, A, 3)
for _ in range(steps):
    net.train(mb)
net.sync()
