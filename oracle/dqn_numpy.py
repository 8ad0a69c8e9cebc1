"""CPU oracle for the network half of the hot path.  TEST INFRASTRUCTURE ONLY.

numpy restatement of /root/reference/src/deepqnetwork.py (DeepQNetwork.__init__
:16-75, _createLayers :77-92, _setInput :94-100, update_target_network :102-105,
train :107-172, predict :174-186) with the Neon semantics A1..A12 listed in
SURVEY.md §8a-bis.  Neon (NervanaSystems/neon, unpinned: README.md:35-37 says
"git clone ... && make" on master, API usage dates it to ~v1.3-1.5) is NOT
vendored and NOT installable here, and the reference ships no tests, golden
vectors or weights for this path.

    *** PARITY UNPINNED at the Neon boundary. ***

What pins this file instead: tests/test_oracle_dqn.py cross-checks forward,
gradients and the RMSProp update against an independent torch-CPU autograd
implementation, and tests/golden/dqn_oracle_golden.npz freezes its outputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (simple_dqn_amd/) never does.

Layouts at this boundary are Neon's (A1, A2):
  W1 (C*R*S=256, K=32), W2 (512, 64), W3 (576, 64): row index c*R*S + r*S + s
  W4 (512, 3136): nin index flattens conv3 output in (K, P, Q) order
  W5 (A, 512)
"""
import numpy as np

# (R, S, K, stride) per conv layer: deepqnetwork.py:83-87
CONV = [(8, 8, 32, 4), (4, 4, 64, 2), (3, 3, 64, 1)]
FC_HIDDEN = 512                                   # :89


def layer_shapes(num_actions, history_length=4, H=84, W=84):
    shapes, C, h, w = [], history_length, H, W
    for (R, S, K, st) in CONV:
        shapes.append((C * R * S, K))
        h, w, C = (h - R) // st + 1, (w - S) // st + 1, K
    shapes.append((FC_HIDDEN, C * h * w))
    shapes.append((num_actions, FC_HIDDEN))
    return shapes


def xavier_weights(num_actions, seed, dtype=np.float32, history_length=4, H=84, W=84):
    """A4: Xavier(local=True) for conv -> fan_in = W.shape[0]; Xavier(local=False)
    for affine -> fan_in = W.shape[1]; U(-k, k), k = sqrt(3 / fan_in)."""
    rng = np.random.RandomState(seed)
    ws = []
    for i, shp in enumerate(layer_shapes(num_actions, history_length, H, W)):
        fan_in = shp[0] if i < 3 else shp[1]
        k = np.sqrt(3.0 / fan_in)
        ws.append(rng.uniform(-k, k, size=shp).astype(dtype))
    return ws


def _im2col(x, R, S, st):
    """x (N, C, H, W) -> cols (N, P*Q, C*R*S) with column index c*R*S + r*S + s (A2)."""
    N, C, H, W = x.shape
    P, Q = (H - R) // st + 1, (W - S) // st + 1
    sn, sc, sh, sw = x.strides
    v = np.lib.stride_tricks.as_strided(
        x, shape=(N, P, Q, C, R, S), strides=(sn, sh * st, sw * st, sc, sh, sw), writeable=False)
    return np.ascontiguousarray(v).reshape(N, P * Q, C * R * S), P, Q


def _col2im(dcols, C, H, W, R, S, st):
    """adjoint of _im2col: dcols (N, P*Q, C*R*S) -> dx (N, C, H, W)."""
    N = dcols.shape[0]
    P, Q = (H - R) // st + 1, (W - S) // st + 1
    d = dcols.reshape(N, P, Q, C, R, S)
    dx = np.zeros((N, C, H, W), dtype=dcols.dtype)
    for r in range(R):
        for s in range(S):
            dx[:, :, r:r + st * P:st, s:s + st * Q:st] += d[:, :, :, :, r, s].transpose(0, 3, 1, 2)
    return dx


class OracleDQN:
    def __init__(self, num_actions, batch_size=32, history_length=4, screen_height=84, screen_width=84,
                 discount_rate=0.99, clip_error=1.0, min_reward=-1.0, max_reward=1.0,
                 learning_rate=0.00025, decay_rate=0.95, epsilon=None, target_steps=10000,
                 dtype=np.float32, weights=None, seed=0, optimizer="rmsprop", beta_1=0.9, beta_2=0.999,
                 half_activations=False, exact_conv1_input=None):
        self.num_actions = num_actions
        self.batch_size = batch_size
        self.history_length = history_length
        self.screen_dim = (screen_height, screen_width)
        self.discount_rate = discount_rate
        self.clip_error = clip_error
        self.min_reward, self.max_reward = min_reward, max_reward
        assert optimizer in ("rmsprop", "adam", "adadelta")               # deepqnetwork.py:50-61
        self.optimizer, self.beta_1, self.beta_2 = optimizer, beta_1, beta_2
        if epsilon is None:                                                # Neon defaults [neon-recalled]
            epsilon = 1e-8 if optimizer == "adam" else 1e-6
        self.lr, self.rho, self.eps = learning_rate, decay_rate, epsilon
        self.dtype = np.dtype(dtype).type
        # --datatype float16 mode of the HIP path (BASELINE.json configs[4]; semantics are OURS — Neon's fp16 backend is
        # GPU-only and unpinned): activations, deltas and the MFMA weight operands are rounded to IEEE half, every
        # accumulation, the master weights, the gradients and the optimizer state stay fp32.
        self.half = bool(half_activations)
        # conv1's input operand in half mode: the exact byte with the 1 / 255 applied to the fp32 sum (the library's default at every batch size
        # since conv1 rides in the float16 forward chain: conv_ssh.h; the weight gradient likewise at B >= 128: c1w_h_kernel<true>), or
        # half(x / 255) (the first forms: problems_h16.h ldh8_u8; still the weight gradient's staging at B < 128 — 2e-4 of that gradient,
        # tools/exp/c1w_h_iso.py — modelled exact here).  None = as the library does by default.
        self.exact_conv1_input = True if exact_conv1_input is None else bool(exact_conv1_input)
        self.loss_scale = 1024.0                                  # deltas are stored as half(delta * 1024) (power of two: exact)
        ws = weights if weights is not None else xavier_weights(num_actions, seed, dtype, history_length,
                                                                screen_height, screen_width)
        self.W = [np.array(w, dtype=dtype) for w in ws]
        self.S = [np.zeros_like(w) for w in self.W]               # RMSProp state / Adam m / Adadelta E[g^2] (init 0)
        self.S2 = [np.zeros_like(w) for w in self.W]              # Adam v / Adadelta E[dx^2]
        # deepqnetwork.py:64-73: separate target model iff target_steps, else alias
        self.target_enabled = bool(target_steps)
        self.Wt = [w.copy() for w in self.W] if self.target_enabled else self.W
        self.train_iterations = 0
        self.callback = None
        self.last_cost = None

    # ---- forward -----------------------------------------------------------
    def _normalize(self, states_u8):                              # _setInput :94-100
        return states_u8.astype(self.dtype) / self.dtype(255)

    def _h(self, x):
        """round to half and back (no-op unless half_activations)"""
        return x.astype(np.float16).astype(self.dtype) if self.half else x

    def _hd(self, d):
        """a delta tensor as the HIP path stores it: half(d * loss_scale), used as (stored / loss_scale)"""
        if not self.half:
            return d
        s = self.dtype(self.loss_scale)
        return (d * s).astype(np.float16).astype(self.dtype) / s

    def fprop(self, W, x, keep=False):
        """x (N, C, H, W) normalised. Returns q (N, A) [+ saved tensors]."""
        if not (self.half and self.exact_conv1_input):
            x = self._h(x)
        acts, cols_all = [x], []
        a = x
        for li, (R, S, K, st) in enumerate(CONV):
            cols, P, Q = _im2col(a, R, S, st)
            if li == 0 and self.half and self.exact_conv1_input:
                # the library's operation order (conv1 in conv_ssh.h / conv1_hb_kernel): exact byte x half weight products summed in fp32, THEN
                # 1 / 255 — (x / 255) . W differs by ~1e-7, which moves ~2e-4 of the half-rounded activations by an ulp and, through
                # flipped Rectlin gates, the gradients by 1e-2 (tools/exp/h16_grad_b32.py)
                z = (_im2col(np.rint(a * self.dtype(255)), R, S, st)[0] @ self._h(W[li])) * self.dtype(1.0 / 255.0)
            else:
                z = cols @ self._h(W[li])                         # (N, PQ, K)
            z = self._h(np.maximum(z, 0))                         # Rectlin (A5)
            a = np.ascontiguousarray(z.transpose(0, 2, 1)).reshape(x.shape[0], K, P, Q)
            acts.append(a)
            cols_all.append(cols)
        # (half mode: conv1's weight gradient reads the same half-rounded normalised frames as its forward pass — every
        #  MFMA operand of the fp16 mode is half; round 1 re-read them in fp32 for its fp32-MFMA wgrad)
        a3f = a.reshape(x.shape[0], -1)                           # (K,P,Q) flatten (A2)
        a4 = np.maximum(a3f @ self._h(W[3]).T, 0)                 # a4 and fc5 stay fp32
        q = a4 @ W[4].T
        if keep:
            return q, (acts, cols_all, a3f, a4)
        return q

    def predict(self, states_u8):                                 # :174-186
        assert states_u8.shape == (self.batch_size, self.history_length) + self.screen_dim
        return self.fprop(self.W, self._normalize(states_u8))

    def update_target_network(self):                              # :102-105
        if self.target_enabled:
            self.Wt = [w.copy() for w in self.W]

    # ---- train -------------------------------------------------------------
    def td_targets(self, preq, maxpostq, actions, rewards, terminals):
        """deepqnetwork.py:133-143: host-side float (=fp64) arithmetic, stored into
        the backend dtype. preq (N, A)."""
        targets = preq.copy()
        rewards = np.clip(rewards, self.min_reward, self.max_reward)
        for i, action in enumerate(actions):
            if terminals[i]:
                targets[i, action] = float(rewards[i])
            else:
                targets[i, action] = float(rewards[i]) + self.discount_rate * float(maxpostq[i])
        return targets

    def gradients(self, minibatch):
        """Everything in train() up to (not including) the optimizer. Returns
        (grads in Neon layout (sum over batch, A8), cost, deltas_clipped, preq)."""
        prestates, actions, rewards, poststates, terminals = minibatch
        assert prestates.shape == poststates.shape and len(prestates.shape) == 4
        N = prestates.shape[0]
        postq = self.fprop(self.Wt, self._normalize(poststates))              # :119-120
        maxpostq = postq.max(axis=1)                                          # :124
        preq, (acts, cols_all, a3f, a4) = self.fprop(self.W, self._normalize(prestates), keep=True)  # :128-129
        targets = self.td_targets(preq, maxpostq, actions, rewards, terminals).astype(self.dtype)
        deltas = preq - targets                                               # get_errors (A7) :149
        cost = self.dtype((0.5 * (deltas * deltas).sum(axis=1)).mean())       # get_cost (A7) :154
        if self.clip_error:                                                   # :158-159
            deltas = np.clip(deltas, -self.clip_error, self.clip_error).astype(self.dtype)
        # ---- bprop (A8) :162
        g = [None] * 5
        g[4] = deltas.T @ a4                                                  # (A, 512)
        d4 = self._hd((deltas @ self.W[4]) * (a4 > 0))
        g[3] = d4.T @ a3f                                                     # (512, 3136)
        d = self._hd((d4 @ self._h(self.W[3])) * (a3f > 0))                   # (N, 3136) in (K,P,Q)
        for li in (2, 1, 0):
            R, S, K, st = CONV[li]
            a_out = acts[li + 1]
            d = d.reshape(a_out.shape)                                        # (N, K, P, Q)
            dmat = np.ascontiguousarray(d.reshape(N, K, -1).transpose(0, 2, 1))   # (N, PQ, K)
            cols = cols_all[li]
            g[li] = np.einsum('nmc,nmk->ck', cols, dmat, optimize=True).astype(self.dtype)
            if li > 0:
                a_in = acts[li]
                dcols = dmat @ self._h(self.W[li]).T                          # (N, PQ, CRS)
                d = self._hd(_col2im(dcols, a_in.shape[1], a_in.shape[2], a_in.shape[3], R, S, st) * (a_in > 0))
        return g, cost, deltas, preq

    def rmsprop(self, grads, batch):
        """A9 + A10, in Neon's operation order [neon-recalled, optimizers.py RMSProp.optimize]:
            grad   = grad / be.bsz
            state  = decay_rate * state + square(grad) * (1.0 - decay_rate)
            param  = param - (grad * learning_rate) / (sqrt(state + epsilon) + epsilon)
        every scalar constant rounded to the backend dtype, one rounding per op (no fma)."""
        t = self.dtype
        for i in range(5):
            gr = (grads[i] / t(batch)).astype(t)
            self.S[i] = (t(self.rho) * self.S[i] + (gr * gr) * t(1.0 - self.rho)).astype(t)
            self.W[i] = (self.W[i] - (gr * t(self.lr)) / (np.sqrt(self.S[i] + t(self.eps)) + t(self.eps))).astype(t)

    def adam(self, grads, batch, epoch):
        """Neon Adam.optimize [neon-recalled]: t = epoch + 1; l = lr*sqrt(1-b2^t)/(1-b1^t);
        m = m*b1 + (1-b1)*g; v = v*b2 + (1-b2)*g*g; param -= l*m / (sqrt(v) + eps)."""
        t = self.dtype
        tt = epoch + 1
        l = self.lr * np.sqrt(1 - self.beta_2 ** tt) / (1 - self.beta_1 ** tt)
        for i in range(5):
            gr = (grads[i] / t(batch)).astype(t)
            self.S[i] = (self.S[i] * t(self.beta_1) + t(1.0 - self.beta_1) * gr).astype(t)
            self.S2[i] = (self.S2[i] * t(self.beta_2) + (t(1.0 - self.beta_2) * gr) * gr).astype(t)
            self.W[i] = (self.W[i] - (t(l) * self.S[i]) / (np.sqrt(self.S2[i]) + t(self.eps))).astype(t)

    def adadelta(self, grads, batch):
        """Neon Adadelta.optimize [neon-recalled]: E[g^2] = d*E[g^2] + (1-d)*g*g; dx = sqrt((E[dx^2]+eps)/(E[g^2]+eps))*g;
        E[dx^2] = d*E[dx^2] + (1-d)*dx*dx; param -= dx."""
        t = self.dtype
        for i in range(5):
            gr = (grads[i] / t(batch)).astype(t)
            self.S[i] = (self.S[i] * t(self.rho) + (t(1.0 - self.rho) * gr) * gr).astype(t)
            upd = (np.sqrt((self.S2[i] + t(self.eps)) / (self.S[i] + t(self.eps))) * gr).astype(t)
            self.S2[i] = (self.S2[i] * t(self.rho) + (t(1.0 - self.rho) * upd) * upd).astype(t)
            self.W[i] = (self.W[i] - upd).astype(t)

    def optimize(self, grads, batch, epoch=0):
        if self.optimizer == "rmsprop":
            self.rmsprop(grads, batch)
        elif self.optimizer == "adam":
            self.adam(grads, batch, epoch)
        else:
            self.adadelta(grads, batch)

    def train(self, minibatch, epoch=0):                                      # :107-172
        grads, cost, _, _ = self.gradients(minibatch)
        self.optimize(grads, minibatch[0].shape[0], epoch)
        self.train_iterations += 1
        self.last_cost = cost
        if self.callback:
            self.callback.on_train(cost)
        return cost
