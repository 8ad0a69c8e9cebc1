import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, simple_dqn_amd as sd
from util import make_args, random_minibatch
from oracle.dqn_numpy import xavier_weights
A, B = 3, 256
mb = random_minibatch(B, A, 396, reward_range=(-2, 3))
def net(opts):
    n = sd.DeepQNetwork(A, make_args(batch_size=B)); n.set_weights(xavier_weights(A, 32), 1); n.set_weights(xavier_weights(A, 31), 0)
    for k, v in opts: n.set_option(k, v)
    n.train(mb); return n
sizes = dict(a2=2 * B * 81 * 64, a3=2 * B * 49 * 64, d3p=B * 121 * 64, d2p=B * 121 * 64, d1=B * 400 * 32)
rel = lambda a, b: float(np.abs(a - b).max() / max(1e-9, np.abs(b).max()))
ref = net([("keep_gradients", 1), ("bt_planes", 0)])
for tag, opts in (("fused", []), ("unfused", [("fused_launches", 0)])):
    n = net([("keep_gradients", 1), ("bt_planes", 9)] + opts)
    print(tag, {k: "%.1e" % rel(n.debug_read(k, s), ref.debug_read(k, s)) for k, s in sizes.items()}, ["%.1e" % rel(n.get_layer(i, 3), ref.get_layer(i, 3)) for i in range(5)])
    x, y = n.debug_read("d2p", sizes["d2p"]).reshape(B, 11, 11, 64), ref.debug_read("d2p", sizes["d2p"]).reshape(B, 11, 11, 64)
    d = np.abs(x - y); idx = np.unravel_index(d.argmax(), d.shape); print("  worst d2p at", idx, x[idx], y[idx], "bad fraction %.4f" % float((d > 1e-5 * np.abs(y).max()).mean()))
    bad = np.argwhere(d > 1e-5 * np.abs(y).max()); print("  bad n range", bad[:, 0].min() if len(bad) else None, bad[:, 0].max() if len(bad) else None, "channels", np.unique(bad[:, 3])[:20] if len(bad) else None)
