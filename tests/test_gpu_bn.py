"""--batch_norm (deepqnetwork.py:26,83-89) on the MI355X against oracle/dqn_bn_numpy.py."""
import numpy as np
import pytest

from oracle.dqn_bn_numpy import OracleDQNBN
from oracle.dqn_numpy import xavier_weights
from util import make_args, random_minibatch

pytestmark = pytest.mark.gpu
Q_TOL = 1e-4


@pytest.fixture(scope="module")
def sd():
    import simple_dqn_amd
    return simple_dqn_amd


def _pair(sd, A, B, seed, randomize=True, **kw):
    args = make_args(batch_size=B, batch_norm=True, **kw)
    net = sd.DeepQNetwork(A, args)
    ws, wt = xavier_weights(A, seed), xavier_weights(A, seed + 1)
    net.set_weights(wt, 1); net.set_weights(ws, 0)
    o = OracleDQNBN(A, batch_size=B, weights=ws, optimizer=args.optimizer)
    o.Wt = [w.copy() for w in wt]
    if randomize:                     # non-trivial BatchNorm parameters and running statistics in both nets
        rng = np.random.RandomState(seed + 7)
        for l in range(4):
            for tgt, (be, ga, gm, gv) in enumerate(((o.beta, o.gamma, o.gmean, o.gvar), (o.beta_t, o.gamma_t, o.gmean_t, o.gvar_t))):
                be[l][:] = rng.uniform(-0.3, 0.3, be[l].shape); ga[l][:] = rng.uniform(0.5, 1.5, ga[l].shape)
                gm[l][:] = rng.uniform(-0.2, 0.2, gm[l].shape); gv[l][:] = rng.uniform(0.5, 2.0, gv[l].shape)
                net.set_bn(l, be[l], ga[l], which=tgt); net.set_bn(l, gm[l], gv[l], which=tgt, running=True)
    return net, o


def _check_state(net, o, tol):
    for i in range(5):
        assert np.abs(net.get_layer(i) - o.W[i]).max() < tol, i
    for l in range(4):
        be, ga = net.get_bn(l); gm, gv = net.get_bn(l, running=True)
        assert np.abs(be - o.beta[l]).max() < tol and np.abs(ga - o.gamma[l]).max() < tol, l
        assert np.abs(gm - o.gmean[l]).max() < tol and np.abs(gv - o.gvar[l]).max() < 10 * tol * max(1.0, np.abs(o.gvar[l]).max()), l


def test_bn_init_and_roundtrip(sd):
    net = sd.DeepQNetwork(4, make_args(batch_size=8, batch_norm=True))
    for which in (0, 1):
        for l, c in enumerate((32, 64, 64, 512)):
            be, ga = net.get_bn(l, which); gm, gv = net.get_bn(l, which, running=True)
            assert np.array_equal(be, np.zeros(c)) and np.array_equal(ga, np.ones(c))          # Neon init
            assert np.array_equal(gm, np.zeros(c)) and np.array_equal(gv, np.zeros(c))
    with pytest.raises(AssertionError):
        sd.DeepQNetwork(4, make_args(batch_size=8)).get_bn(0)


@pytest.mark.parametrize("A,B", [(4, 32), (6, 7)])
def test_bn_predict_parity(sd, A, B):
    net, o = _pair(sd, A, B, 11)
    st = random_minibatch(B, A, 12)[0]
    q, qo = net.predict(st), o.predict(st)
    print("bn predict max abs err %.3e (|Q| max %.3f)" % (np.abs(q - qo).max(), np.abs(qo).max()))
    assert np.abs(q - qo).max() < Q_TOL
    assert np.abs(net.predict_one(st[0]) - qo[0]).max() < Q_TOL          # inference statistics: independent of the batch


@pytest.mark.parametrize("A,B,optimizer", [(4, 32, "rmsprop"), (6, 16, "adam"), (3, 7, "adadelta")])
def test_bn_train_step_parity(sd, A, B, optimizer):
    net, o = _pair(sd, A, B, 21, optimizer=optimizer)
    net.set_option("keep_gradients", 1)
    mb = random_minibatch(B, A, 22)
    g, cost, deltas, preq = o.gradients(mb)
    gm_after = [m.copy() for m in o.gmean]; gv_after = [v.copy() for v in o.gvar]
    o.optimize(g, B)
    net.train(mb)
    q, _ = net.last_q()
    assert np.abs(q - preq).max() < Q_TOL
    for i in range(5):
        gi = net.get_layer(i, 3)
        assert np.abs(gi - g[i]).max() < 5e-4 * max(1e-3, np.abs(g[i]).max()), i
    gb, gg = o._bn_grads
    for l in range(4):
        b_, g_ = net.get_bn(l, 3)
        assert np.abs(b_ - gb[l]).max() < 5e-4 * max(1e-3, np.abs(gb[l]).max()), l
        assert np.abs(g_ - gg[l]).max() < 5e-4 * max(1e-3, np.abs(gg[l]).max()), l
    _check_state(net, o, 2e-5)


def _force(net, o):
    """teacher forcing: the oracle's complete learner state into the HIP net"""
    for i in range(5):
        net.set_layer(i, o.W[i], 0); net.set_layer(i, o.Wt[i], 1); net.set_layer(i, o.S[i], 2)
    for l in range(4):
        net.set_bn(l, o.beta[l], o.gamma[l]); net.set_bn(l, o.beta_t[l], o.gamma_t[l], which=1)
        net.set_bn(l, o.Sb[l], o.Sg[l], which=2)
        net.set_bn(l, o.gmean[l], o.gvar[l], running=True); net.set_bn(l, o.gmean_t[l], o.gvar_t[l], which=1, running=True)


def test_bn_training_from_neon_init_and_target_sync(sd):
    """From Neon's init (beta 0, gamma 1, running statistics 0) — where the first inference passes divide by
    sqrt(~0.1 var) and |Q| is O(10..50) — two free-running steps, then teacher-forced steps across a target sync
    (free-running trajectories of two fp32 implementations separate chaotically)."""
    A, B = 4, 32
    net, o = _pair(sd, A, B, 31, randomize=False)
    held = random_minibatch(B, A, 99)[0]
    for s in range(8):
        if s >= 2:
            _force(net, o)
        if s == 4:
            net.update_target_network(); o.update_target_network()
            for l in range(4):                                         # weights AND running statistics travel (keep_states, :103-105)
                assert all(np.array_equal(a, b) for a, b in zip(net.get_bn(l, which=1, running=True), net.get_bn(l, running=True)))
                assert all(np.array_equal(a, b) for a, b in zip(net.get_bn(l, which=1), net.get_bn(l)))
        mb = random_minibatch(B, A, 32 + s)
        net.train(mb); o.train(mb)
        qo = o.predict(held)
        err = np.abs(net.predict(held) - qo).max()
        print("bn step %d: predict max abs err %.3e (|Q| max %.2f)" % (s + 1, err, np.abs(qo).max()))
        assert err < 2e-4 * max(1.0, np.abs(qo).max()), s
    _check_state(net, o, 1e-4)


def test_bn_replay_path_and_snapshot(sd, tmp_path):
    import ctypes as C
    from bench import fill_ring
    A, B = 4, 32
    args = make_args(batch_size=B, batch_norm=True)
    mem = sd.ReplayMemory(3000, args); fill_ring(mem, 3, A)
    net, _ = _pair(sd, A, B, 41)
    net2, _ = _pair(sd, A, B, 41)
    lib = sd.load()
    mt = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt, 5)
    mt2 = (C.c_uint32 * 625)(); lib.sdqn_mt_seed(mt2, 5)
    c1 = net.train_from_memory(mem, 12, mt_state=mt, want_cost=True)
    costs = []
    for _ in range(12):                                               # same stream, one library call per step + the fused and unfused launch modes
        net2.set_option("fused_launches", len(costs) % 2)
        costs.append(net2.train_from_memory(mem, 1, mt_state=mt2, want_cost=True))
    assert abs(c1 - float(np.mean(costs))) < 1e-6 * max(1.0, abs(c1))
    for i in range(5):
        assert np.array_equal(net.get_layer(i), net2.get_layer(i)), i
    for l in range(4):
        assert all(np.array_equal(a, b) for a, b in zip(net.get_bn(l), net2.get_bn(l)))
        assert all(np.array_equal(a, b) for a, b in zip(net.get_bn(l, running=True), net2.get_bn(l, running=True)))
    p = str(tmp_path / "bn.npz")
    net.save_weights(p)
    net3 = sd.DeepQNetwork(A, args); net3.load_weights(p)
    st = random_minibatch(B, A, 42)[0]
    assert np.array_equal(net.predict(st), net3.predict(st))
    mb = random_minibatch(B, A, 43)
    net.train(mb); net3.train(mb)
    for i in range(5):
        assert np.array_equal(net.get_layer(i), net3.get_layer(i)), i


@pytest.mark.parametrize("overlap", [0, 1])
def test_bn_data_parallel_single_rank(sd, overlap):
    """the data-parallel split (reduce -> all-reduce -> apply; BatchNorm gradients ride in the flat buffer) with a
    1-rank RCCL communicator reproduces the single-GPU batch_norm step bit for bit"""
    from simple_dqn_amd.deepqnetwork import dp_unique_id
    A, B = 4, 32
    n1, _ = _pair(sd, A, B, 61)
    n2, _ = _pair(sd, A, B, 61)
    n2.set_option("dp_overlap", overlap)
    n2.dp_init(dp_unique_id(), 0, 1)
    for s in range(4):
        mb = random_minibatch(B, A, 62 + s)
        n1.train(mb); n2.train(mb)
    for i in range(5):
        assert np.array_equal(n1.get_layer(i), n2.get_layer(i)), i
    for l in range(4):
        assert all(np.array_equal(a, b) for a, b in zip(n1.get_bn(l), n2.get_bn(l)))
        assert all(np.array_equal(a, b) for a, b in zip(n1.get_bn(l, running=True), n2.get_bn(l, running=True)))
        assert all(np.array_equal(a, b) for a, b in zip(n1.get_bn(l, 2), n2.get_bn(l, 2)))
    n2.dp_shutdown()
