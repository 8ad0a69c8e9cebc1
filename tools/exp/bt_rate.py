"""Experiment helper (not product): step rate of the fused train loop at B >= 128 under option sets.
   python tools/exp/bt_rate.py "opt=v,opt=v" "opt=v" ...     (one rate line per argument; "" = defaults)   env: B, A, STEPS, REPS, DATATYPE"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args
from bench import fill_ring
B, A = int(os.environ.get("B", 256)), int(os.environ.get("A", 3))
STEPS, REPS = int(os.environ.get("STEPS", 400)), int(os.environ.get("REPS", 3))
args = make_args(batch_size=B, datatype=os.environ.get("DATATYPE", "float32"))
mem = sd.ReplayMemory(100000, args); fill_ring(mem, 1, A)
for spec in (sys.argv[1:] or [""]):
    net = sd.DeepQNetwork(A, args); net.update_target_network()
    for kv in [x for x in spec.split(",") if x]:
        k, v = kv.split("="); net.set_option(k, int(v))
    mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 5)
    net.train_from_memory(mem, 100, mt_state=mt, want_cost=False); net.sync()
    r = []
    for _ in range(REPS):
        t = time.perf_counter(); net.train_from_memory(mem, STEPS, mt_state=mt, want_cost=False); net.sync()
        r.append(STEPS / (time.perf_counter() - t))
    print("%-60s %s max %d steps/s = %.1f us/step" % (spec or "(defaults)", [round(x) for x in r], max(r), 1e6 / max(r)), flush=True)
    del net
