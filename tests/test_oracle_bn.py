"""The --batch_norm oracle (oracle/dqn_bn_numpy.py) against torch autograd with functional batch_norm."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.dqn_bn_numpy import BN_EPS, OracleDQNBN  # noqa: E402
from oracle.dqn_numpy import xavier_weights  # noqa: E402
from util import random_minibatch  # noqa: E402

torch = pytest.importorskip("torch")
F = torch.nn.functional


def _torch_net(ws, o, x, train, dt):
    shapes = [(32, 4, 8, 8), (64, 32, 4, 4), (64, 64, 3, 3)]
    P = [torch.tensor(w.T.reshape(sh).copy(), dtype=dt, requires_grad=True) for w, sh in zip(ws[:3], shapes)] + \
        [torch.tensor(ws[3].copy(), dtype=dt, requires_grad=True), torch.tensor(ws[4].copy(), dtype=dt, requires_grad=True)]
    beta = [torch.tensor(b.copy(), dtype=dt, requires_grad=True) for b in o.beta]
    gamma = [torch.tensor(g.copy(), dtype=dt, requires_grad=True) for g in o.gamma]
    rm = [torch.tensor(m.copy(), dtype=dt) for m in o.gmean]; rv = [torch.tensor(v.copy(), dtype=dt) for v in o.gvar]
    h = x
    for l, st in enumerate((4, 2, 1)):
        h = F.conv2d(h, P[l], stride=st)
        h = F.relu(F.batch_norm(h, rm[l].clone(), rv[l].clone(), gamma[l], beta[l], training=train, momentum=0.1, eps=BN_EPS))
    h = h.flatten(1) @ P[3].t()
    h = F.relu(F.batch_norm(h, rm[3].clone(), rv[3].clone(), gamma[3], beta[3], training=train, momentum=0.1, eps=BN_EPS))
    return h @ P[4].t(), P, beta, gamma


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 2e-4)])
def test_bn_forward_backward_matches_torch(dtype, tol):
    A, B = 4, 8
    ws = xavier_weights(A, 3, dtype)
    o = OracleDQNBN(A, batch_size=B, weights=ws, dtype=dtype)
    rng = np.random.RandomState(1)
    for l in range(4):                                           # non-trivial BN parameters and running statistics
        o.beta[l][:] = rng.uniform(-0.3, 0.3, o.beta[l].shape); o.gamma[l][:] = rng.uniform(0.5, 1.5, o.gamma[l].shape)
        o.gmean[l][:] = rng.uniform(-0.2, 0.2, o.gmean[l].shape); o.gvar[l][:] = rng.uniform(0.5, 2.0, o.gvar[l].shape)
    o.update_target_network()
    mb = random_minibatch(B, A, 2)
    dt = torch.float64 if dtype == np.float64 else torch.float32
    x = torch.tensor(o._normalize(mb[0]), dtype=dt)
    # inference mode
    q_inf, *_ = _torch_net(ws, o, x, False, dt)
    assert np.abs(o.predict(mb[0]) - q_inf.detach().numpy()).max() < tol
    # training mode forward + gradients of sum(q * c)
    gm0 = [m.copy() for m in o.gmean]
    q, (acts, cols_all, a3f, a4, sv) = o.fprop_bn(o.W, o._normalize(mb[0]), inference=False, keep=True)
    qt, P, beta, gamma = _torch_net(ws, o.__class__(A, batch_size=B, weights=ws, dtype=dtype) if False else _restore(o, gm0), x, True, dt)
    assert np.abs(q - qt.detach().numpy()).max() < tol
    # full gradient check through gradients(): same TD machinery as the parent, BN in the middle
    o2 = _restore(o, gm0)
    g, cost, deltas, preq = o2.gradients(mb)
    qt.backward(torch.tensor(deltas, dtype=dt))
    for i in range(5):
        gt = P[i].grad.numpy()
        gt = gt.reshape(gt.shape[0], -1).T if i < 3 else gt
        assert np.abs(g[i] - gt).max() < tol * max(1.0, np.abs(gt).max()), i
    gb, gg = o2._bn_grads
    for l in range(4):
        assert np.abs(gb[l] - beta[l].grad.numpy()).max() < tol * max(1.0, np.abs(gb[l]).max()), l
        assert np.abs(gg[l] - gamma[l].grad.numpy()).max() < tol * max(1.0, np.abs(gg[l]).max()), l


def _restore(o, gm0):
    """the training-mode forward updates the running means: rewind them so two passes see the same state"""
    for l in range(4):
        o.gmean[l][:] = gm0[l]
    return o


def test_bn_running_statistics_and_target_copy():
    A, B = 4, 8
    o = OracleDQNBN(A, batch_size=B, weights=xavier_weights(A, 5))
    mb = random_minibatch(B, A, 6)
    o.train(mb)
    assert all(np.abs(m).max() > 0 for m in o.gmean) and all((v > 0).all() for v in o.gvar)
    assert all(np.array_equal(m, np.zeros_like(m)) for m in o.gmean_t)          # target untouched until the sync
    q_before = o.fprop_bn(o.Wt, o._normalize(mb[3]), inference=True, target=True)
    o.update_target_network()
    assert all(np.array_equal(a, b) for a, b in zip(o.gmean, o.gmean_t))
    assert not np.allclose(q_before, o.fprop_bn(o.Wt, o._normalize(mb[3]), inference=True, target=True))
    assert all(not np.array_equal(b, np.zeros_like(b)) for b in o.beta)         # the optimizer moved beta / gamma
