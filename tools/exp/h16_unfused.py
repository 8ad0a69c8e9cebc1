import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, simple_dqn_amd as sd
from util import make_args, random_minibatch
B, A = int(os.environ.get("B", 256)), 3
mb = random_minibatch(B, A, 3)
for spec in sys.argv[1:] or ["fused_launches=0"]:
    net = sd.DeepQNetwork(A, make_args(batch_size=B, datatype=os.environ.get("DATATYPE", "float16"))); net.update_target_network()
    for kv in [x for x in spec.split(",") if x]:
        k, v = kv.split("="); net.set_option(k, int(v))
    for _ in range(5): net.train(mb)
    net.profile(True, -1); net.profile_reset()
    for _ in range(30): net.train(mb)
    print(spec, {p["name"][:14]: round(p["total_ms"] / p["launches"] * 1e3, 1) for p in net.profile_read() if p["launches"]})
