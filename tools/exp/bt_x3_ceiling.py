import os, sys, time, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, simple_dqn_amd as sd
from util import make_args, random_minibatch
B, A = 256, 3
NAMES = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "fc4_fwd", 4: "head", 5: "fc4_dgrad", 12: "update", 16: "bwd3", 17: "bwd2", 18: "bwd1"}
mb = random_minibatch(B, A, 3)
for x in (0, 9, 19, 16):
    net = sd.DeepQNetwork(A, make_args(batch_size=B)); net.update_target_network()
    net.set_option("bt_x", x); net.set_option("bt:3", 1); net.set_option("bt:5", 1); net.set_option("s4", 7)
    for _ in range(5): net.train(mb)
    net.profile(True, -1); net.profile_reset()
    for _ in range(30): net.train(mb)
    us = {p["id"]: p["total_ms"] / p["launches"] * 1e3 for p in net.profile_read() if p["launches"]}
    print("bt_x=%2d total %.1f | %s" % (x, sum(v for k, v in us.items() if k in NAMES), "  ".join("%s %.1f" % (NAMES[k], us[k]) for k in sorted(us) if k in NAMES)), flush=True)
