"""Experiment helper (not product): per-launch times and step rate of the single-rank data-parallel step (serial form) next to the
single-GPU step, same process.  env: B, A"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from simple_dqn_amd.deepqnetwork import dp_unique_id
from util import make_args
from bench import fill_ring
B, A = int(os.environ.get("B", 32)), int(os.environ.get("A", 4))
args = make_args(batch_size=B)
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
for dp in (0, 1, 0, 1):
    net = sd.DeepQNetwork(A, args); net.update_target_network()
    if dp:
        net.set_option("dp_overlap", 0)
        net.dp_init(dp_unique_id(), 0, 1)
    mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
    net.train_from_memory(mem, 300, mt_state=mt, want_cost=False); net.sync()
    r = []
    for _ in range(3):
        t = time.perf_counter(); net.train_from_memory(mem, 2000, mt_state=mt, want_cost=False); net.sync()
        r.append(2000 / (time.perf_counter() - t))
    net.profile(True, -1); net.profile_reset()
    net.train_from_memory(mem, 200, mt_state=mt, want_cost=False)
    us = {p["name"]: round(p["total_ms"] / p["launches"] * 1e3, 2) for p in net.profile_read() if p["launches"]}
    n_l = {p["name"]: p["launches"] for p in net.profile_read() if p["launches"]}
    net.profile(False)
    print("dp=%d  %d steps/s (%.1f us)  | " % (dp, max(r), 1e6 / max(r)) + "  ".join("%s %.2f x%d" % (k, v, n_l[k] // 200) for k, v in us.items()), flush=True)
    if dp: net.dp_shutdown()
    del net
