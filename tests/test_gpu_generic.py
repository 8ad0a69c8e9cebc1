"""The configurations the tuned kernels do not cover — `--datatype float64` (src/main.py:53) and screens / history lengths
other than 84 x 84 x 4 (src/main.py:27-28,34) — run on the library's generic im2col + GEMM path (csrc/generic_net.hip)
behind the same C ABI.  Oracle: oracle/dqn_numpy.py (already geometry- and dtype-generic), oracle/replay_numpy.py.
float64 is held to 1e-9 (summation order only); float32 to the contract's 1e-4 on Q with gradients at 2e-5 relative."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle.dqn_numpy import OracleDQN, xavier_weights, layer_shapes
from oracle.replay_numpy import ReplayOracle, synthetic_fill
from util import make_args

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    import simple_dqn_amd
    return simple_dqn_amd


def _minibatch(B, A, hist, H, W, seed, p_term=0.25):
    rng = np.random.RandomState(seed)
    pre = rng.randint(0, 256, (B, hist, H, W), dtype=np.uint8)
    post = rng.randint(0, 256, (B, hist, H, W), dtype=np.uint8)
    return pre, rng.randint(0, A, B).astype(np.uint8), rng.randint(-2, 3, B).astype(np.int64), post, rng.rand(B) < p_term


def _pair(sd, A, B, hist, H, W, dtype, seed, **kw):
    args = make_args(batch_size=B, history_length=hist, screen_height=H, screen_width=W, datatype=dtype, **kw)
    net = sd.DeepQNetwork(A, args)
    npd = np.float64 if dtype == "float64" else np.float32
    ws = xavier_weights(A, seed, npd, hist, H, W)
    wt = xavier_weights(A, seed + 1, npd, hist, H, W)
    net.set_weights(wt, 1)
    net.set_weights(ws, 0)
    o = OracleDQN(A, batch_size=B, history_length=hist, screen_height=H, screen_width=W, dtype=npd, weights=ws,
                  optimizer=kw.get("optimizer", "rmsprop"), learning_rate=kw.get("learning_rate", 0.00025))
    o.Wt = [w.copy() for w in wt]
    return net, o


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))


@pytest.mark.parametrize("dtype,geom", [("float64", (4, 84, 84)), ("float64", (3, 60, 52)), ("float32", (2, 64, 48)), ("float32", (5, 36, 36))])
def test_one_step_gradients_update_and_q(sd, dtype, geom):
    """One train step: gradient sums of every layer, cost, updated weights + RMSProp state, Q of a held-out batch."""
    hist, H, W = geom
    A, B = 6, 7
    net, o = _pair(sd, A, B, hist, H, W, dtype, 11)
    assert [w.shape for w in net.get_weights()] == layer_shapes(A, hist, H, W)
    mb = _minibatch(B, A, hist, H, W, 5)
    held = _minibatch(B, A, hist, H, W, 6)[0]
    tol_g, tol_w, tol_q = (1e-11, 1e-12, 1e-10) if dtype == "float64" else (2e-5, 2e-6, 1e-5)
    q0, oq0 = net.predict(held), o.predict(held)
    assert q0.dtype == (np.float64 if dtype == "float64" else np.float32)
    assert np.abs(q0 - oq0).max() < tol_q
    costs = []
    net.callback = type("CB", (), {"on_train": lambda self, c: costs.append(c)})()
    g_o, cost_o, _, _ = o.gradients(mb)
    net.train(mb, 0)
    o.train(mb, 0)
    assert abs(costs[0] - float(cost_o)) <= 1e-6 * max(1.0, abs(float(cost_o)))
    for l in range(5):
        assert _rel(net.get_layer(l, 3), g_o[l]) < tol_g, ("gradient", l)
        assert _rel(net.get_layer(l, 0), o.W[l]) < tol_w, ("weights", l)
        assert _rel(net.get_layer(l, 2), o.S[l]) < max(tol_g * 4, 1e-10), ("rmsprop state", l)
        assert np.array_equal(net.get_layer(l, 1), o.Wt[l].astype(net.get_layer(l, 1).dtype)), ("target untouched", l)
    assert np.abs(net.predict(held) - o.predict(held)).max() < tol_q


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_free_running_steps_and_target_sync(sd, dtype):
    """Ten free-running steps on fresh minibatches with a target sync in the middle.  float64 leaves no room for gate flips to matter
    (differences ~1e-13 cannot move an activation across zero unless it sits there), so the whole trajectory is held to 1e-9;
    float32 to the contract's 1e-4 on the Q-values of a held-out batch."""
    hist, H, W, A, B = 4, 84, 84, 4, 8
    if dtype == "float32":
        hist, H, W = 3, 52, 68                       # (84 x 84 x 4 float32 is the tuned path: tests/test_gpu_dqn.py)
    net, o = _pair(sd, A, B, hist, H, W, dtype, 21)
    held = _minibatch(B, A, hist, H, W, 99)[0]
    for i in range(10):
        mb = _minibatch(B, A, hist, H, W, 100 + i)
        if i == 5:
            net.update_target_network(); o.update_target_network()
        net.train(mb, 0); o.train(mb, 0)
    err = np.abs(net.predict(held) - o.predict(held)).max()
    assert err < (1e-9 if dtype == "float64" else 1e-4), err
    assert net.train_iterations == 10


@pytest.mark.parametrize("optimizer", ["adam", "adadelta"])
def test_float64_other_optimizers(sd, optimizer):
    hist, H, W, A, B = 2, 44, 40, 3, 4
    net, o = _pair(sd, A, B, hist, H, W, "float64", 31, optimizer=optimizer)
    for i in range(3):
        mb = _minibatch(B, A, hist, H, W, 200 + i)
        net.train(mb, epoch=i); o.train(mb, epoch=i)
    for l in range(5):
        assert _rel(net.get_layer(l, 0), o.W[l]) < 1e-11, (optimizer, l)
        assert _rel(net.get_layer(l, 2), o.S[l]) < 1e-9 and _rel(net.get_layer(l, 4), o.S2[l]) < 1e-9, (optimizer, l)


def test_replay_gather_and_fused_loop_on_another_geometry(sd):
    """ReplayMemory with 60 x 52 screens and history_length 3: sampled indexes + gathered bytes bit-exact against the oracle ring
    (replay_memory.py:54-79), then train_from_memory (native sampler -> device gather -> step) against the oracle fed its own
    getMinibatch() from the same random stream, float64."""
    hist, H, W, A, B, size = 3, 60, 52, 5, 6, 900
    args = make_args(batch_size=B, history_length=hist, screen_height=H, screen_width=W, datatype="float64")
    mem = sd.ReplayMemory(size, args)
    omem = ReplayOracle(size, H, W, hist, B)
    synthetic_fill(mem, 77, num_actions=A)
    synthetic_fill(omem, 77, num_actions=A)
    mem.sync_mirror()
    random.seed(4)
    got = [x.copy() for x in mem.getMinibatch()]
    random.seed(4)
    want = omem.getMinibatch()
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert got[0].shape == (B, hist, H, W)
    net, o = _pair(sd, A, B, hist, H, W, "float64", 41)
    random.seed(9)
    cost = net.train_from_memory(mem, 4, want_cost=True)
    state_after = random.getstate()
    random.seed(9)
    costs = [float(o.train(tuple(x.copy() for x in omem.getMinibatch()))) for _ in range(4)]
    assert random.getstate() == state_after                     # the native sampler consumed exactly the reference's draws
    assert abs(cost - np.mean(costs)) < 1e-6 * max(1.0, abs(np.mean(costs)))
    for l in range(5):
        assert _rel(net.get_layer(l, 0), o.W[l]) < 1e-11, l
    # ... and a direct write through the tracked views reaches the mirror for this geometry too
    mem.screens[10:40] ^= 0x5A
    omem.screens[10:40] ^= 0x5A
    random.seed(12); got = [x.copy() for x in mem.getMinibatch()]
    random.seed(12); want = omem.getMinibatch()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_acting_path_on_another_geometry(sd):
    """predict_one / DeviceStateBuffer + predict_state (agent.py:55-61) against predict() of the padded batch."""
    hist, H, W, A, B = 3, 48, 56, 4, 4
    args = make_args(batch_size=B, history_length=hist, screen_height=H, screen_width=W)
    net, o = _pair(sd, A, B, hist, H, W, "float32", 51)
    rng = np.random.RandomState(3)
    buf = sd.DeviceStateBuffer(args)
    for _ in range(70):                                        # > 64 adds: the device ring of the buffer wraps
        buf.add(rng.randint(0, 256, (H, W), dtype=np.uint8))
    state = buf.getState()
    batch = np.zeros((B, hist, H, W), np.uint8); batch[0] = state
    q = net.predict(batch)[0]
    assert np.abs(q - o.predict(batch)[0]).max() < 1e-5
    assert np.array_equal(net.predict_one(state), q)
    assert np.array_equal(net.predict_state(buf), q)


def test_what_stays_refused_says_so(sd):
    with pytest.raises(NotImplementedError):
        sd.DeepQNetwork(4, make_args(batch_size=4, screen_height=64, datatype="float16"))
    with pytest.raises(NotImplementedError):
        sd.DeepQNetwork(4, make_args(batch_size=4, history_length=2, batch_norm=True))
    with pytest.raises(AssertionError):                         # 20 x 20 screens do not survive conv2 (deepqnetwork.py:85)
        sd.DeepQNetwork(4, make_args(batch_size=4, screen_height=20, screen_width=20))
    mem = sd.ReplayMemory(200, make_args(batch_size=4, screen_height=60, screen_width=52, history_length=3))
    synthetic_fill(mem, 1, num_actions=4); mem.sync_mirror()
    with pytest.raises(AssertionError):                         # a 60 x 52 x 3 ring cannot feed an 84 x 84 x 4 network
        sd.DeepQNetwork(4, make_args(batch_size=4)).train_from_memory(mem, 1)
    net = sd.DeepQNetwork(4, make_args(batch_size=4, datatype="float64"))
    with pytest.raises(AssertionError):
        net.train_from_memory(mem, 1)
    with pytest.raises((RuntimeError, AssertionError)):
        net.apply_update(8)
    with pytest.raises((RuntimeError, AssertionError)):
        net.set_option("grad_only", 1)


def test_hundred_free_running_steps_against_the_float64_yardstick(sd):
    """the 1e-4 contract is met on the MAE of teacher-forced steps, while free-running float32 implementations of this
    algorithm separate through ReLU / clip gate flips.  The honest question for a free-running trajectory is therefore not 'HIP == oracle'
    but 'is HIP float32 as close to the exact trajectory as the reference-style float32 arithmetic is?'.  The float64 network on the GPU
    (generic path, itself held to the float64 oracle at 1e-9 over ten free-running steps above) is that exact trajectory, cheap enough for
    100 steps: the tuned float32 path, the numpy float32 oracle and the float64 network train on the same 100 minibatches (same sampled
    indexes, target sync every 25 steps) from the same weights; on a held-out batch the tuned path's distance from float64 must not
    exceed 1.5 x the oracle's (or the 1e-4 contract)."""
    A, B, size, steps = 4, 32, 6000, 100
    a32, a64 = make_args(batch_size=B), make_args(batch_size=B, datatype="float64")
    mem = sd.ReplayMemory(size, a32)
    synthetic_fill(mem, 5150, num_actions=A)
    mem.sync_mirror()
    ws, wt = xavier_weights(A, 5151), xavier_weights(A, 5152)
    n32, n64 = sd.DeepQNetwork(A, a32), sd.DeepQNetwork(A, a64)
    for n in (n32, n64):
        n.set_weights(wt, 1); n.set_weights(ws, 0)
    o32 = OracleDQN(A, batch_size=B, weights=ws)
    o32.Wt = [w.copy() for w in wt]
    held = _minibatch(B, A, 4, 84, 84, 5153)[0]
    random.seed(5154)
    worst = []
    for i in range(steps):
        if i and i % 25 == 0:
            n32.update_target_network(); n64.update_target_network(); o32.update_target_network()
        idx = mem.sample_indexes().copy()
        mb = tuple(np.array(x) for x in mem.gather(idx))
        n32.train_indexes(mem, idx); n64.train_indexes(mem, idx); o32.train(mb)
        if i % 10 == 9:
            q64 = n64.predict(held)
            worst.append((float(np.abs(n32.predict(held) - q64).max()), float(np.abs(o32.predict(held) - q64).max())))
    d_hip, d_orc = worst[-1]
    print("free-running distance from float64 every 10 steps (hip fp32, oracle fp32):", ["%.1e/%.1e" % w for w in worst])
    assert d_hip <= max(1.5 * d_orc, 1e-4), worst
    assert max(w[0] for w in worst) <= max(2.0 * max(w[1] for w in worst), 1e-4), worst
