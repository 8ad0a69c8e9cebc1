import os, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = 32, 4
args = make_args(batch_size=B)
mem = sd.ReplayMemory(50000, args); fill_ring(mem, 1, A)
for hoist in (0, 1):
    net = sd.DeepQNetwork(A, args); net.update_target_network(); net.set_option("hoist", hoist)
    mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
    net.train_from_memory(mem, 100, mt_state=mt, want_cost=False); net.sync()
    net.profile(True, -1); net.profile_reset()
    net.train_from_memory(mem, 400, mt_state=mt, want_cost=False)
    r = {p["name"].split("(")[0]: round(p["total_ms"] / p["launches"] * 1e3, 2) for p in net.profile_read() if p["launches"] >= 300}
    net.profile(False)
    print("hoist", hoist, r, "sum %.1f" % sum(r.values()))
