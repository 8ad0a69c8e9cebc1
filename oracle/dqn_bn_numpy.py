"""CPU oracle of the --batch_norm variant of the network (deepqnetwork.py:26,83-89).  TEST INFRASTRUCTURE ONLY.

`Conv(..., batch_norm=True)` / `Affine(nout=512, ..., batch_norm=True)` put a Neon BatchNorm layer between the
linear part and the Rectlin of conv1..3 and fc4 (the output Affine has none, :91).  Neon is unpinned and absent
(SURVEY.md §8c), so these semantics are [neon-recalled] (neon/layers/layer.py, class BatchNorm, v1.x):

    rho = 0.9, eps = 1e-3; beta = 0, gamma = 1, gmean = gvar = 0 at init
    x viewed as (features, everything else): per FEATURE MAP for conv (over N*P*Q), per unit for Affine (over N)
    training fprop : xmean = mean(x), xvar = var(x) (biased); gmean = gmean*rho + (1-rho)*xmean (same for gvar);
                     xhat = (x - xmean) / sqrt(xvar + eps);  y = xhat*gamma + beta
    inference fprop: xhat = (x - gmean) / sqrt(gvar + eps);  y = xhat*gamma + beta
    bprop          : grad_gamma = sum(xhat*err); grad_beta = sum(err);
                     dx = gamma * (err - (xhat*grad_gamma + grad_beta)/m) / sqrt(xvar + eps),  m = elements per feature
    the optimizer treats (beta, gamma) like any parameter: grad / be.bsz first (A9), own state each.

Call sites that fix the modes: target net `fprop(inference=True)` (deepqnetwork.py:120), online net in train
`fprop(inference=False)` (:129), predict `fprop(inference=True)` (:180); `update_target_network` copies weights AND
states (keep_states=True, :103-105) — the running statistics travel with it.

    *** PARITY UNPINNED (as for oracle/dqn_numpy.py) ***; tests/test_oracle_bn.py checks the forward/backward math
    against torch.nn.functional.batch_norm + autograd.
"""
import numpy as np

from .dqn_numpy import CONV, OracleDQN, _col2im, _im2col

BN_RHO, BN_EPS = 0.9, 1e-3
BN_FEATURES = [32, 64, 64, 512]


class OracleDQNBN(OracleDQN):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        assert not self.half, "batch_norm with float16 is not defined"
        t = self.dtype
        self.beta = [np.zeros(c, t) for c in BN_FEATURES]
        self.gamma = [np.ones(c, t) for c in BN_FEATURES]
        self.gmean = [np.zeros(c, t) for c in BN_FEATURES]
        self.gvar = [np.zeros(c, t) for c in BN_FEATURES]
        self.Sb = [np.zeros(c, t) for c in BN_FEATURES]; self.Sg = [np.zeros(c, t) for c in BN_FEATURES]
        self.S2b = [np.zeros(c, t) for c in BN_FEATURES]; self.S2g = [np.zeros(c, t) for c in BN_FEATURES]
        self._copy_target()

    def _copy_target(self):
        if self.target_enabled:
            self.beta_t = [b.copy() for b in self.beta]; self.gamma_t = [g.copy() for g in self.gamma]
            self.gmean_t = [m.copy() for m in self.gmean]; self.gvar_t = [v.copy() for v in self.gvar]
        else:
            self.beta_t, self.gamma_t, self.gmean_t, self.gvar_t = self.beta, self.gamma, self.gmean, self.gvar

    def update_target_network(self):
        super().update_target_network()
        self._copy_target()

    # x2d: (rows, features).  Returns y, and (xhat, rstd) when training
    def _bn(self, x2d, l, inference, target=False):
        t = self.dtype
        beta, gamma = (self.beta_t, self.gamma_t) if target else (self.beta, self.gamma)
        if inference:
            gm, gv = (self.gmean_t, self.gvar_t) if target else (self.gmean, self.gvar)
            rstd = (t(1) / np.sqrt(gv[l] + t(BN_EPS))).astype(t)
            xhat = ((x2d - gm[l]) * rstd).astype(t)
            return (xhat * gamma[l] + beta[l]).astype(t), None
        xmean = x2d.mean(axis=0, dtype=np.float64).astype(t)
        xvar = x2d.var(axis=0, dtype=np.float64).astype(t)
        self.gmean[l] = (self.gmean[l] * t(BN_RHO) + t(1 - BN_RHO) * xmean).astype(t)
        self.gvar[l] = (self.gvar[l] * t(BN_RHO) + t(1 - BN_RHO) * xvar).astype(t)
        rstd = (t(1) / np.sqrt(xvar + t(BN_EPS))).astype(t)
        xhat = ((x2d - xmean) * rstd).astype(t)
        return (xhat * gamma[l] + beta[l]).astype(t), (xhat, rstd)

    def fprop_bn(self, W, x, inference, target=False, keep=False):
        N = x.shape[0]
        acts, cols_all, bn_saved = [x], [], []
        a = x
        for li, (R, S, K, st) in enumerate(CONV):
            cols, P, Q = _im2col(a, R, S, st)
            z = (cols @ W[li]).reshape(N * P * Q, K)                     # rows (n, p, q), features K
            y, sv = self._bn(z, li, inference, target)
            y = np.maximum(y, 0).reshape(N, P * Q, K)
            a = np.ascontiguousarray(y.transpose(0, 2, 1)).reshape(N, K, P, Q)
            acts.append(a); cols_all.append(cols); bn_saved.append(sv)
        a3f = a.reshape(N, -1)
        y4, sv4 = self._bn(a3f @ W[3].T, 3, inference, target)
        bn_saved.append(sv4)
        a4 = np.maximum(y4, 0)
        q = a4 @ W[4].T
        if keep:
            return q, (acts, cols_all, a3f, a4, bn_saved)
        return q

    def fprop(self, W, x, keep=False):          # generic entry of the parent: inference mode of the matching net
        return self.fprop_bn(W, x, inference=True, target=(W is self.Wt and W is not self.W), keep=keep)

    def predict(self, states_u8):                                 # :174-186, inference=True
        assert states_u8.shape == (self.batch_size, self.history_length) + self.screen_dim
        return self.fprop_bn(self.W, self._normalize(states_u8), inference=True)

    def _bn_bprop(self, err2d, l, saved):
        t = self.dtype
        xhat, rstd = saved
        m = t(err2d.shape[0])
        gg = (xhat * err2d).sum(axis=0, dtype=np.float64).astype(t)
        gb = err2d.sum(axis=0, dtype=np.float64).astype(t)
        dx = (self.gamma[l] * (err2d - (xhat * gg + gb) / m) * rstd).astype(t)
        return dx, gb, gg

    def gradients(self, minibatch):
        prestates, actions, rewards, poststates, terminals = minibatch
        N = prestates.shape[0]
        postq = self.fprop_bn(self.Wt, self._normalize(poststates), inference=True, target=self.target_enabled)   # :119-120
        maxpostq = postq.max(axis=1)
        preq, (acts, cols_all, a3f, a4, sv) = self.fprop_bn(self.W, self._normalize(prestates), inference=False, keep=True)  # :128-129
        targets = self.td_targets(preq, maxpostq, actions, rewards, terminals).astype(self.dtype)
        deltas = preq - targets
        cost = self.dtype((0.5 * (deltas * deltas).sum(axis=1)).mean())
        if self.clip_error:
            deltas = np.clip(deltas, -self.clip_error, self.clip_error).astype(self.dtype)
        g = [None] * 5
        gbeta, ggamma = [None] * 4, [None] * 4
        g[4] = deltas.T @ a4
        e4 = (deltas @ self.W[4]) * (a4 > 0)                                  # error at BN4's output
        d4, gbeta[3], ggamma[3] = self._bn_bprop(e4, 3, sv[3])
        g[3] = d4.T @ a3f
        d = (d4 @ self.W[3]) * (a3f > 0)                                      # (N, 3136) in (K,P,Q): error at BN3's output
        for li in (2, 1, 0):
            R, S, K, st = CONV[li]
            a_out = acts[li + 1]
            d = d.reshape(a_out.shape)
            e = np.ascontiguousarray(d.reshape(N, K, -1).transpose(0, 2, 1)).reshape(-1, K)   # rows (n,p,q)
            dx, gbeta[li], ggamma[li] = self._bn_bprop(e, li, sv[li])
            dmat = dx.reshape(N, -1, K)
            cols = cols_all[li]
            g[li] = np.einsum('nmc,nmk->ck', cols, dmat, optimize=True).astype(self.dtype)
            if li > 0:
                a_in = acts[li]
                dcols = dmat @ self.W[li].T
                d = _col2im(dcols, a_in.shape[1], a_in.shape[2], a_in.shape[3], R, S, st) * (a_in > 0)
        self._bn_grads = (gbeta, ggamma)
        return g, cost, deltas, preq

    def _step(self, p, gsum, s, s2, batch, epoch):
        t = self.dtype
        gr = (gsum / t(batch)).astype(t)
        if self.optimizer == "rmsprop":
            s[...] = (t(self.rho) * s + (gr * gr) * t(1.0 - self.rho)).astype(t)
            p[...] = (p - (gr * t(self.lr)) / (np.sqrt(s + t(self.eps)) + t(self.eps))).astype(t)
        elif self.optimizer == "adam":
            tt = epoch + 1
            l = self.lr * np.sqrt(1 - self.beta_2 ** tt) / (1 - self.beta_1 ** tt)
            s[...] = (s * t(self.beta_1) + t(1.0 - self.beta_1) * gr).astype(t)
            s2[...] = (s2 * t(self.beta_2) + (t(1.0 - self.beta_2) * gr) * gr).astype(t)
            p[...] = (p - (t(l) * s) / (np.sqrt(s2) + t(self.eps))).astype(t)
        else:
            s[...] = (s * t(self.rho) + (t(1.0 - self.rho) * gr) * gr).astype(t)
            upd = (np.sqrt((s2 + t(self.eps)) / (s + t(self.eps))) * gr).astype(t)
            s2[...] = (s2 * t(self.rho) + (t(1.0 - self.rho) * upd) * upd).astype(t)
            p[...] = (p - upd).astype(t)

    def optimize(self, grads, batch, epoch=0):
        super().optimize(grads, batch, epoch)
        gbeta, ggamma = self._bn_grads
        for l in range(4):
            self._step(self.beta[l], gbeta[l], self.Sb[l], self.S2b[l], batch, epoch)
            self._step(self.gamma[l], ggamma[l], self.Sg[l], self.S2g[l], batch, epoch)
