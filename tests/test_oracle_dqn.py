"""Checks oracle/dqn_numpy.py (the Neon-semantics restatement; parity unpinned at the Neon boundary)
against an independent torch-CPU autograd implementation and freezes its outputs in a golden file."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.dqn_numpy import OracleDQN, xavier_weights, layer_shapes
from util import random_minibatch


def _torch_weights(W, dt):
    conv = lambda w, C, R, K: torch.tensor(w.reshape(C, R, R, K).transpose(3, 0, 1, 2).copy(), dtype=dt, requires_grad=True)
    return [conv(W[0], 4, 8, 32), conv(W[1], 32, 4, 64), conv(W[2], 64, 3, 64),
            torch.tensor(W[3].copy(), dtype=dt, requires_grad=True), torch.tensor(W[4].copy(), dtype=dt, requires_grad=True)]


def _torch_fwd(w, x, dt):
    x = torch.tensor(x).to(dt) / 255
    x = F.relu(F.conv2d(x, w[0], stride=4))
    x = F.relu(F.conv2d(x, w[1], stride=2))
    x = F.relu(F.conv2d(x, w[2], stride=1))
    x = F.relu(x.reshape(x.shape[0], -1) @ w[3].T)
    return x @ w[4].T


def _neon_grads(w):
    g = [w[i].grad.numpy() for i in range(5)]
    return [g[0].transpose(1, 2, 3, 0).reshape(256, 32), g[1].transpose(1, 2, 3, 0).reshape(512, 64),
            g[2].transpose(1, 2, 3, 0).reshape(576, 64), g[3], g[4]]


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-11), (np.float32, 3e-5)])
def test_gradients_match_torch_autograd(dt, tol):
    torch.set_num_threads(4)
    A, B = 6, 6
    o = OracleDQN(A, batch_size=B, dtype=dt, weights=xavier_weights(A, 1, dt))
    o.Wt = [w.copy() for w in xavier_weights(A, 2, dt)]
    mb = random_minibatch(B, A, 3)
    pre, act, rew, post, term = mb
    g, cost, deltas, preq = o.gradients(mb)
    tdt = torch.float64 if dt == np.float64 else torch.float32
    w, wt = _torch_weights(o.W, tdt), _torch_weights(o.Wt, tdt)
    with torch.no_grad():
        mq = _torch_fwd(wt, post, tdt).max(1).values
    q = _torch_fwd(w, pre, tdt)
    r = torch.tensor(np.clip(rew, -1, 1)).to(tdt)
    y = torch.where(torch.tensor(term), r, r + 0.99 * mq)
    qa = q.gather(1, torch.tensor(act.astype(np.int64))[:, None])[:, 0]
    d = (qa - y).detach()
    (d.clamp(-1, 1) * qa).sum().backward()          # dL/dq = clipped error on the taken action only
    assert np.abs(q.detach().numpy() - preq).max() < tol
    assert abs(float((0.5 * d * d).mean()) - float(cost)) < tol * 10
    for a, b in zip(_neon_grads(w), g):
        assert np.abs(a - b).max() < tol * max(1.0, np.abs(b).max())


def test_rmsprop_formula():
    A, B = 4, 4
    o = OracleDQN(A, batch_size=B, dtype=np.float64, weights=xavier_weights(A, 1, np.float64))
    mb = random_minibatch(B, A, 5)
    W0 = [w.copy() for w in o.W]
    g, _, _, _ = o.gradients(mb)
    o.train(mb)
    for i in range(5):
        gr = g[i] / B
        s = 0.05 * gr * gr                               # state starts at 0 (A10)
        exp = W0[i] - 0.00025 * gr / (np.sqrt(s + 1e-6) + 1e-6)
        assert np.abs(exp - o.W[i]).max() < 1e-12
    assert o.train_iterations == 1


def test_td_target_semantics():
    # terminal -> y = clip(r); clip_error bounds the delta; cost is pre-clip (deepqnetwork.py:136-159)
    A, B = 3, 5
    o = OracleDQN(A, batch_size=B, dtype=np.float64, weights=xavier_weights(A, 7, np.float64))
    pre, act, rew, post, term = random_minibatch(B, A, 9, reward_range=(-5, 6))
    term[:] = True
    g, cost, deltas, preq = o.gradients((pre, act, rew, post, term))
    y = np.clip(rew, -1, 1).astype(np.float64)
    d = preq[np.arange(B), act] - y
    assert abs(cost - (0.5 * d * d).mean()) < 1e-12
    assert np.allclose(deltas[np.arange(B), act], np.clip(d, -1, 1))
    assert np.count_nonzero(deltas) <= B


def test_predict_zero_rows():
    # no biases + ReLU(0)=0 -> all-zero states give exactly 0 Q-values
    A, B = 4, 3
    o = OracleDQN(A, batch_size=B, weights=xavier_weights(A, 1))
    st = np.zeros((B, 4, 84, 84), np.uint8)
    st[0] = np.random.RandomState(0).randint(0, 256, (4, 84, 84))
    q = o.predict(st)
    assert np.all(q[1:] == 0) and np.any(q[0] != 0)


def test_golden_frozen(golden_dir):
    path = os.path.join(golden_dir, "dqn_oracle_golden.npz")
    A, B = 4, 4
    o = OracleDQN(A, batch_size=B, weights=xavier_weights(A, 11))
    mb = random_minibatch(B, A, 12)
    hold = random_minibatch(B, A, 13)[0]
    q0 = o.predict(hold)
    costs = [float(o.train(mb)) for _ in range(3)]
    q3 = o.predict(hold)
    if not os.path.exists(path):                      # first run writes the fixture (committed)
        np.savez(path, q0=q0, q3=q3, costs=np.array(costs))
    f = np.load(path)
    assert np.abs(f["q0"] - q0).max() < 1e-5 and np.abs(f["q3"] - q3).max() < 1e-4
    assert np.abs(f["costs"] - np.array(costs)).max() < 1e-5


def test_layer_shapes():
    assert layer_shapes(4) == [(256, 32), (512, 64), (576, 64), (512, 3136), (4, 512)]
    assert sum(a * b for a, b in layer_shapes(4)) == 1685504
