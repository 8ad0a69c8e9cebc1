"""DeepQNetwork — drop-in for /root/reference/src/deepqnetwork.py:15-192, no Neon.

Same constructor (num_actions, args), methods (train / predict / update_target_network /
load_weights / save_weights) and attributes (train_iterations, callback, batch_size, num_actions).
The whole step — target forward, online forward, TD target, clipped error, backward, RMSProp —
runs in libsdqn_hip.so on the device with zero host round trips (the reference does six, :119-171).
"""
import array
import ctypes as C
import logging

import numpy as np

from . import _lib
from ._lazy import LazyMinibatchArray

logger = logging.getLogger(__name__)

# (rows, cols) in Neon layout per layer (deepqnetwork.py:83-91): conv (C*R*S, K), affine (nout, nin);
# 84x84x4 inputs give (256, 32), (512, 64), (576, 64), (512, 3136), (A, 512)
def layer_shapes(num_actions, history_length=4, screen_height=84, screen_width=84):
    shapes, c, h, w = [], history_length, screen_height, screen_width
    for (r, s_, k, st) in ((8, 8, 32, 4), (4, 4, 64, 2), (3, 3, 64, 1)):
        assert h >= r and w >= s_, "screen too small for the layer stack of deepqnetwork.py:83-87"
        shapes.append((c * r * s_, k))
        h, w, c = (h - r) // st + 1, (w - s_) // st + 1, k
    return shapes + [(512, c * h * w), (num_actions, 512)]


class DeepQNetwork:
    def __init__(self, num_actions, args):
        self._lib = _lib.load()
        self.num_actions = num_actions
        self.batch_size = args.batch_size
        self.discount_rate = args.discount_rate
        self.history_length = args.history_length
        self.screen_dim = (args.screen_height, args.screen_width)
        self._state_shape = (self.batch_size, self.history_length) + self.screen_dim      # what train() / predict() take
        self.clip_error = args.clip_error
        self.min_reward = args.min_reward
        self.max_reward = args.max_reward
        self.batch_norm = bool(getattr(args, "batch_norm", False))                      # :26
        if getattr(args, "backend", "hip") == "cpu":
            raise NotImplementedError("there is no CPU backend: libsdqn_hip is MI355X-only")
        dt = str(getattr(args, "datatype", "float32"))
        if dt not in ("float32", "float16", "float64"):                                 # main.py:53
            raise NotImplementedError("datatype %s: float16, float32 and float64 are implemented" % dt)
        self.datatype = dt
        # float64 and screens other than 84x84x4 run on the library's generic im2col + GEMM path (csrc/generic_net.hip), everything
        # else on the tuned kernels; the choice is the library's, from the configuration alone
        self._f64 = dt == "float64"
        self._np = np.float64 if self._f64 else np.float32
        self._shapes = layer_shapes(num_actions, self.history_length, *self.screen_dim)
        tuned_geom = (self.history_length,) + tuple(self.screen_dim) == (4, 84, 84)
        if not tuned_geom and (dt == "float16" or self.batch_norm):
            raise NotImplementedError("float16 and batch_norm are implemented for 84x84 screens with history_length 4")
        if getattr(args, "stochastic_round", False):                                    # main.py:54 -> gen_backend(stochastic_round=...)
            raise NotImplementedError("stochastic rounding (a Neon fp16 GPU-backend feature) is not implemented")
        if self.batch_norm and dt != "float32":
            raise NotImplementedError("batch_norm is implemented for float32 only")
        optimizer = getattr(args, "optimizer", "rmsprop")
        assert optimizer in ("rmsprop", "adam", "adadelta"), "Unknown optimizer"      # deepqnetwork.py:61
        self.optimizer = optimizer
        _lib.bind_device(args)                                                        # :29-34 gen_backend(device_id=args.device_id)
        cfg = _lib.NetCfg()
        cfg.batch_size, cfg.history_length = self.batch_size, self.history_length
        cfg.screen_height, cfg.screen_width = self.screen_dim
        cfg.num_actions = num_actions
        cfg.target_enabled = 1 if getattr(args, "target_steps", 10000) else 0          # :64
        cfg.discount_rate, cfg.clip_error = float(self.discount_rate), float(self.clip_error or 0)
        cfg.min_reward, cfg.max_reward = float(self.min_reward), float(self.max_reward)
        cfg.learning_rate = float(getattr(args, "learning_rate", 0.00025))
        cfg.decay_rate = float(getattr(args, "decay_rate", 0.95))
        cfg.optimizer = ("rmsprop", "adam", "adadelta").index(optimizer)              # :50-59
        cfg.datatype = {"float32": 0, "float16": 1, "float64": 2}[dt]                # :33
        cfg.loss_scale = float(getattr(args, "loss_scale", 1024.0))
        cfg.batch_norm = 1.0 if self.batch_norm else 0.0                              # :83-89 Conv/Affine(batch_norm=...)
        # Neon's defaults (the reference passes none): RMSProp/Adadelta epsilon 1e-6, Adam epsilon 1e-8, betas 0.9/0.999
        cfg.epsilon = float(getattr(args, "optimizer_epsilon", 1e-8 if optimizer == "adam" else 1e-6))
        cfg.beta_1, cfg.beta_2 = float(getattr(args, "beta_1", 0.9)), float(getattr(args, "beta_2", 0.999))
        h = C.c_void_p()
        _lib.check(self._lib.sdqn_net_create(C.byref(h), C.byref(cfg)))
        self._h = h
        import os
        for env, opt in (("SDQN_FUSED_LAUNCHES", b"fused_launches"), ("SDQN_XCD_MAP", b"xcd_map"), ("SDQN_H16_WGRAD_MFMA", b"h16_wgrad_mfma")):
            if os.environ.get(env) is not None:                       # A/B switches for benchmarking
                _lib.check(self._lib.sdqn_net_set_option(h, opt, int(os.environ[env])))
        if os.environ.get("SDQN_F4_SHARE"):                           # tuning: "s3,s2" percent of fc4-wgrad tiles in bwd3 / bwd2
            s3, s2 = [int(x) for x in os.environ["SDQN_F4_SHARE"].split(",")]
            _lib.check(self._lib.sdqn_net_set_option(h, b"f4_share3", s3))
            _lib.check(self._lib.sdqn_net_set_option(h, b"f4_share2", s2))
        self._mt_buf = (C.c_uint32 * _lib.MT_WORDS)()
        self._act_out = C.c_int(); self._act_greedy = self._lib.sdqn_net_act_greedy
        self.train_iterations = 0
        self.callback = None
        self.save_weights_prefix = getattr(args, "save_weights_prefix", None)
        # Xavier init (A4): online layers first, then the target model's own draw (:65-70)
        # (gen_backend(rng_seed=args.random_seed), :31: seed 0 is a seed here, unlike main.py:89's `if args.random_seed`)
        rng = np.random.RandomState(getattr(args, "random_seed", None))
        for which in ((0, 1) if cfg.target_enabled else (0,)):
            for i, shp in enumerate(self._shapes):
                fan_in = shp[0] if i < 3 else shp[1]
                k = np.sqrt(3.0 / fan_in)
                self.set_layer(i, rng.uniform(-k, k, size=shp).astype(self._np), which)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and self._lib is not None:
            self._lib.sdqn_net_destroy(h)

    # ---- weights at the boundary are ALWAYS in Neon layout --------------------------------------
    def set_layer(self, layer, w, which=0):
        w = np.ascontiguousarray(w, dtype=self._np)
        assert w.shape == self._shapes[layer], (w.shape, layer)
        if self._f64:
            _lib.check(self._lib.sdqn_net_set_weights_f64(self._h, which, layer, _lib.ptr(w, C.c_double), w.size))
        else:
            _lib.check(self._lib.sdqn_net_set_weights(self._h, which, layer, _lib.ptr(w, C.c_float), w.size))

    def get_layer(self, layer, which=0):
        w = np.empty(self._shapes[layer], dtype=self._np)
        if self._f64:
            _lib.check(self._lib.sdqn_net_get_weights_f64(self._h, which, layer, _lib.ptr(w, C.c_double), w.size))
        else:
            _lib.check(self._lib.sdqn_net_get_weights(self._h, which, layer, _lib.ptr(w, C.c_float), w.size))
        return w

    # --batch_norm: BatchNorm layer l = 0..3 (after conv1, conv2, conv3, fc4).  which as above for (beta, gamma);
    # running=True addresses (gmean, gvar) of the online (which=0) or target (which=1) net
    def get_bn(self, l, which=0, running=False):
        assert self.batch_norm and 0 <= l < 4
        c = (32, 64, 64, 512)[l]
        w = np.empty((2, c), dtype=np.float32)
        _lib.check(self._lib.sdqn_net_get_weights(self._h, (5 + which) if running else which, 5 + l, _lib.ptr(w, C.c_float), w.size))
        return w[0].copy(), w[1].copy()

    def set_bn(self, l, first, second, which=0, running=False):
        assert self.batch_norm and 0 <= l < 4
        w = np.ascontiguousarray(np.stack([first, second]), dtype=np.float32)
        assert w.shape == (2, (32, 64, 64, 512)[l])
        _lib.check(self._lib.sdqn_net_set_weights(self._h, (5 + which) if running else which, 5 + l, _lib.ptr(w, C.c_float), w.size))

    def set_weights(self, weights, which=0):
        for i, w in enumerate(weights):
            self.set_layer(i, w, which)

    def get_weights(self, which=0):
        return [self.get_layer(i, which) for i in range(5)]

    # ---- reference API ----------------------------------------------------------------------------
    def update_target_network(self):                               # :102-105
        _lib.check(self._lib.sdqn_net_update_target(self._h))

    def train(self, minibatch, epoch=0):                           # :107-172
        prestates, actions, rewards, poststates, terminals = minibatch
        ps, qs, as_, rs, ts = prestates.shape, poststates.shape, actions.shape, rewards.shape, terminals.shape     # (the reference's asserts, :108-115)
        assert len(ps) == 4
        assert len(qs) == 4
        assert len(as_) == 1
        assert len(rs) == 1
        assert len(ts) == 1
        assert ps == qs
        assert ps[0] == as_[0] == rs[0] == qs[0] == ts[0]
        assert ps == self._state_shape
        act = np.ascontiguousarray(actions, dtype=np.uint8)
        rew = np.ascontiguousarray(rewards, dtype=np.int64)
        term = np.ascontiguousarray(terminals).astype(np.uint8)
        cost = C.c_float()
        want = self.callback is not None
        if self.optimizer == "adam":
            _lib.check(self._lib.sdqn_net_set_epoch(self._h, int(epoch)))        # optimizer.optimize(.., epoch), :165
        # The reference's loop body net.train(mem.getMinibatch()): the gathered states are still on the device.
        #  * untouched lazy views of one ReplayMemory (nobody has looked at or written the host buffers since the last gather): the step
        #    reads the device copy, the 1.8 MB never cross PCIe in either direction;
        #  * the memory's own buffers, fetched but not written since: same, declared the round-3 way;
        #  * anything else (written buffers, foreign arrays): uploaded.
        mem = None
        if (isinstance(prestates, LazyMinibatchArray) and isinstance(poststates, LazyMinibatchArray) and prestates._mem is poststates._mem
                and prestates._which == "pre" and poststates._which == "post"):
            mem = prestates._mem
            if mem._mb_pending:
                # (generation 2^64 - 1 = "the minibatch of the last gather() call, not fetched yet": the library refuses — RuntimeError —
                #  if anything has overwritten the device minibatch since; everything in this package that re-gathers into it
                #  materialises a pending minibatch first, so this is the backstop for foreign callers of the C ABI)
                _lib.check(self._lib.sdqn_replay_declare_minibatch_on_device(mem._h, 0xFFFFFFFFFFFFFFFF))
                pre_p, post_p = mem._mb_ptrs                                   # (addresses only: they name the handle's buffers)
            else:
                prestates, poststates = mem._states("pre"), mem._states("post")   # fetch; then the ordinary path below
                mem = None
        if mem is None:
            pre = np.ascontiguousarray(prestates, dtype=np.uint8)
            post = np.ascontiguousarray(poststates, dtype=np.uint8)
            pre_p, post_p = _lib.ptr(pre, C.c_uint8), _lib.ptr(post, C.c_uint8)
            owner = getattr(prestates, "_owner", None)
            m2 = owner() if owner is not None else None
            if (m2 is not None and prestates is getattr(m2, "_prestates", None) and poststates is getattr(m2, "_poststates", None)
                    and not getattr(m2, "_mb_pending", True) and not getattr(m2, "_mb_dirty", True)):
                _lib.check(self._lib.sdqn_replay_declare_minibatch_clean(m2._h))
        _lib.check(self._lib.sdqn_net_train_host(self._h, pre_p, _lib.ptr(act, C.c_uint8), _lib.ptr(rew, C.c_int64), post_p,
                                                 _lib.ptr(term, C.c_uint8), C.byref(cost) if want else None))
        self.train_iterations += 1                                 # :168
        if self.callback:
            self.callback.on_train(cost.value)                     # :171-172

    def predict(self, states):                                     # :174-186
        assert states.shape == ((self.batch_size, self.history_length,) + self.screen_dim)
        st = np.ascontiguousarray(states, dtype=np.uint8)
        q = np.empty((self.batch_size, self.num_actions), dtype=self._np)
        if self._f64:
            _lib.check(self._lib.sdqn_net_predict_f64(self._h, _lib.ptr(st, C.c_uint8), _lib.ptr(q, C.c_double)))
        else:
            _lib.check(self._lib.sdqn_net_predict(self._h, _lib.ptr(st, C.c_uint8), _lib.ptr(q, C.c_float)))
        return q

    def predict_one(self, state):
        """Acting-path fast path: Q-values of one state u8[hist,H,W] -> float32[A]; identical to
        predict(padded_batch)[0] (agent.py:55-61) without computing the zero rows."""
        assert state.shape == (self.history_length,) + self.screen_dim
        st = np.ascontiguousarray(state, dtype=np.uint8)
        q = np.empty((self.num_actions,), dtype=np.float32)
        _lib.check(self._lib.sdqn_net_predict_one(self._h, _lib.ptr(st, C.c_uint8), _lib.ptr(q, C.c_float)))
        return q

    def predict_state(self, state_buffer):
        """Acting path with a DeviceStateBuffer: Q-values float32[A] of the buffered state, read in place from HBM
        (no state upload; same numbers as predict(state_buffer.getStateMinibatch())[0])."""
        q = np.empty((self.num_actions,), dtype=np.float32)
        _lib.check(self._lib.sdqn_net_predict_state(self._h, state_buffer._h, _lib.ptr(q, C.c_float)))
        return q

    def act_greedy(self, state_buffer):
        """agent.py:55-59 in one call: int(np.argmax(predict_state(state_buffer)))."""
        a = self._act_out
        _lib.check(self._act_greedy(self._h, state_buffer._h, C.byref(a), None))
        return a.value

    def act_step(self, state_buffer, mem, screen, action=0, reward=0, terminal=False, speculate=False):
        """One environment transition in ONE library call: `state_buffer.add(screen)` and — with a device-backed replay memory —
        `mem.add(action, reward, screen, terminal)` (agent.py:62 / replay_memory.py:26-34); `speculate` also enqueues the acting forward of
        the new state, so that the next predict_state(state_buffer) only collects Q-values that are already on their way."""
        assert screen.shape == state_buffer.dims
        assert mem is None or screen.shape == mem.dims              # (replay_memory.py:28: mem.add's own assert)
        scr = np.ascontiguousarray(screen, dtype=np.uint8)
        rh = mem._h if mem is not None else None
        _lib.check(self._lib.sdqn_net_act_step(self._h, state_buffer._h, rh, _lib.ptr(scr, C.c_uint8), int(action), int(reward),
                                               int(bool(terminal)), int(bool(speculate))))

    def load_weights(self, load_path):                             # :188-189
        """Own .npz snapshots, or a Neon pickle (the reference's `model.load_params`, best effort: neon_compat.py)."""
        if not str(load_path).endswith(".npz"):
            from .neon_compat import read_neon_pickle
            ws, states, A = read_neon_pickle(load_path)
            assert A == self.num_actions, "snapshot has %d actions, the network %d" % (A, self.num_actions)
            for i in range(5):
                self.set_layer(i, ws[i], 0)
                if states is not None and self.optimizer == "rmsprop":
                    self.set_layer(i, states[i], 2)
            return                                                 # (online model only, like model.load_params: the target
                                                                   #  net follows at the next update_target_network, agent.py:105)
        with np.load(load_path) as f:
            for which, key in ((0, "W"), (1, "Wt"), (2, "S"), (4, "S2")):
                for i in range(5):
                    name = "%s%d" % (key, i)
                    if name in f:
                        self.set_layer(i, f[name], which)
                if self.batch_norm:
                    for l in range(4):
                        if "%s_bn%d" % (key, l) in f:
                            self.set_bn(l, *f["%s_bn%d" % (key, l)], which=which)
            if self.batch_norm:
                for which, key in ((0, "run"), (1, "run_t")):
                    for l in range(4):
                        if "%s_bn%d" % (key, l) in f:
                            self.set_bn(l, *f["%s_bn%d" % (key, l)], which=which, running=True)
            if "train_iterations" in f:
                self.train_iterations = int(f["train_iterations"])

    def save_weights(self, save_path):                             # :191-192
        """.npz (everything needed to resume: online + target weights, optimizer state, step counter) or, for a
        path ending in .prm / .pkl, a Neon-style pickle of the online model (neon_compat.py)."""
        if str(save_path).endswith((".prm", ".pkl")):
            from .neon_compat import write_neon_pickle
            write_neon_pickle(save_path, [self.get_layer(i, 0) for i in range(5)],
                              [self.get_layer(i, 2) for i in range(5)] if self.optimizer == "rmsprop" else None)
            return
        d = {}
        for which, key in ((0, "W"), (1, "Wt"), (2, "S")) + ((() if self.optimizer == "rmsprop" else ((4, "S2"),))):
            for i in range(5):
                d["%s%d" % (key, i)] = self.get_layer(i, which)
            if self.batch_norm:
                for l in range(4):
                    d["%s_bn%d" % (key, l)] = np.stack(self.get_bn(l, which))
        if self.batch_norm:
            for which, key in ((0, "run"), (1, "run_t")):
                for l in range(4):
                    d["%s_bn%d" % (key, l)] = np.stack(self.get_bn(l, which, running=True))
        d["train_iterations"] = np.int64(self.train_iterations)
        with open(save_path, "wb") as f:
            np.savez(f, **d)

    # ---- additive fast paths (SURVEY.md §8b "new, additive") ------------------------------------------
    def set_epoch(self, epoch):
        _lib.check(self._lib.sdqn_net_set_epoch(self._h, int(epoch)))

    def train_from_memory(self, mem, n_steps=1, use_global_random=True, mt_state=None, want_cost=None):
        """n_steps x { sample ; fused gather+train } entirely inside the library (agent.py:108-114's
        loop body).  Consumes Python's global random stream unless an explicit 625-word mt_state
        (ctypes uint32 array) is given."""
        import random
        mem._check_mirror()
        self._keep_pending_minibatch(mem)
        if mt_state is None:
            # Python's generator state goes in as a copy (array('I') of the 625 words: 6 us; a ctypes slice assignment took 25); afterwards
            # Python's own generator is advanced by exactly the 32-bit words the library drew — one getrandbits call instead of
            # rebuilding a 625-tuple and random.setstate (another 40 us per call)
            st = random.getstate()
            arr = array.array("I", st[1])
            mt = (C.c_uint32 * _lib.MT_WORDS).from_buffer(arr)
            w0 = C.c_uint64(); self._lib.sdqn_mt_words(C.byref(w0))
        else:
            mt = mt_state

        def resync():
            if mt_state is None:
                w1 = C.c_uint64(); self._lib.sdqn_mt_words(C.byref(w1))
                n = w1.value - w0.value
                if n:
                    random.getrandbits(32 * n)                     # ceil(k / 32) genrand_uint32 calls: the same n words (Modules/_randommodule.c)

        want = (self.callback is not None) if want_cost is None else want_cost
        cost = C.c_float()
        if self.callback is not None and n_steps > 1:
            # the reference reports every step to the callback (deepqnetwork.py:168-172: train_iterations, then
            # on_train(cost)): step by step, so Statistics' running mean sees the same sequence; no callback -> one call
            total = 0.0
            try:
                for _ in range(int(n_steps)):
                    _lib.check(self._lib.sdqn_net_train_many(self._h, mem._h, mt, 1, C.byref(cost)))
                    self.train_iterations += 1
                    self.callback.on_train(cost.value)
                    total += cost.value
            finally:
                resync()
            return total / n_steps if want else None
        if self.callback is not None and n_steps == 1 and want_cost is None and hasattr(self.callback, "on_train_deferred"):
            # the callback can take the cost later: enqueue the step, hand it a collector (deepqnetwork.py:168-172, order kept)
            ticket = C.c_int64()
            try:
                _lib.check(self._lib.sdqn_net_train_many_deferred(self._h, mem._h, mt, 1, C.byref(ticket)))
            finally:
                resync()
            self.train_iterations += 1
            tk = ticket.value

            def collect(tk=tk):
                c = C.c_float()
                _lib.check(self._lib.sdqn_net_cost_collect(self._h, tk, C.byref(c)))
                return c.value
            self.callback.on_train_deferred(collect, self.train_iterations)
            return None
        try:
            _lib.check(self._lib.sdqn_net_train_many(self._h, mem._h, mt, int(n_steps), C.byref(cost) if want else None))
        finally:
            resync()
        self.train_iterations += n_steps
        if self.callback and n_steps:
            self.callback.on_train(cost.value)
        return cost.value if want else None

    def _keep_pending_minibatch(self, mem):
        """A getMinibatch() whose states nobody has looked at yet lives ONLY in the device minibatch.  The generic path (float64, other
        screen geometries) gathers a fused step's states into that very buffer: fetch the pending one to the host first, so that a
        later net.train(mb) / np.asarray(mb[0]) still sees its own states (the tuned 84x84x4 path gathers inside conv1 and never
        touches the buffer)."""
        if getattr(mem, "_mb_pending", False) and self.step_structure()[0] == "generic":
            mem._materialize()

    def train_indexes(self, mem, indexes, want_cost=False):
        idx = np.ascontiguousarray(indexes, dtype=np.int64)
        mem._check_mirror()
        self._keep_pending_minibatch(mem)
        cost = C.c_float()
        _lib.check(self._lib.sdqn_net_train_replay(self._h, mem._h, _lib.ptr(idx, C.c_int64), C.byref(cost) if want_cost else None))
        self.train_iterations += 1
        return cost.value if want_cost else None

    def set_option(self, name, value):
        """Tuning / test hooks of the library (sdqn_net_set_option), e.g. 'keep_gradients' (materialise the fc4 gradient; disables the
        fused fc4 RMSProp), 'fused_launches', 'bt:<kernel id>' (throughput-regime menu), 'nw:<id>', 'tps:<layer>', 's4'.  Retired
        experiments ('two_streams', ...) are refused by name."""
        _lib.check(self._lib.sdqn_net_set_option(self._h, name.encode(), int(value)))
        if name == "dp_overlap":
            self._dp_overlap_opt = int(value)

    def sync(self):
        _lib.check(self._lib.sdqn_net_sync(self._h))

    def last_q(self):
        preq = np.empty((self.batch_size, self.num_actions), dtype=np.float32)
        mq = np.empty((self.batch_size,), dtype=np.float32)
        _lib.check(self._lib.sdqn_net_last_q(self._h, _lib.ptr(preq, C.c_float), _lib.ptr(mq, C.c_float)))
        return preq, mq

    def debug_read(self, name, n):
        out = np.empty(int(n), dtype=np.float32)
        _lib.check(self._lib.sdqn_net_debug_read(self._h, name.encode(), _lib.ptr(out, C.c_float), out.size))
        return out

    # ---- per-kernel device timing (bench.py roofline leg) -------------------------------------------
    def profile(self, enable, kernel=-1):
        _lib.check(self._lib.sdqn_net_profile(self._h, int(bool(enable)), int(kernel)))

    def profile_reset(self):
        _lib.check(self._lib.sdqn_net_profile_reset(self._h))

    def profile_read(self):
        n = C.c_int()
        _lib.check(self._lib.sdqn_net_profile_count(C.byref(n)))
        out = []
        for k in range(n.value):
            name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
            _lib.check(self._lib.sdqn_net_profile_read(self._h, k, C.byref(name), C.byref(ms), C.byref(cnt)))
            out.append(dict(id=k, name=name.value.decode(), total_ms=ms.value, launches=cnt.value))
        return out

    # ---- data parallel -----------------------------------------------------------------------------
    def dp_init(self, unique_id, rank, nranks, rccl=None, vote=None, probe_timeout_ms=5000, inject_probe_timeout=False):
        """One learner per GPU: RCCL communicator over `nranks` ranks.  With option "dp_overlap" at its default (-1, auto) and
        nranks >= 2 the OVERLAPPED form (fc4's 95 % of the gradient all-reduced + applied on a second communicator under the rest of
        the step) is tried by rule: every rank probes it with bounded waits (sdqn_dp_probe), the ranks agree through `vote` — a
        callable taking this rank's bool and returning the AND over all ranks, e.g. a gloo all-reduce(MIN) of the control plane — and
        all of them activate it, or all of them tear the second communicator down and run the serial form (one all-reduce on the
        library stream).  Without a `vote` the ranks cannot agree, so the serial form runs.  dp_form() says which one it is."""
        path = (rccl or _lib.rccl_path()).encode()
        # Every rank must issue the SAME control-plane collectives whatever happens locally: the ranks whose start-up went fine will sit
        # in the vote, so a rank whose init / probe raised casts vote(False) before it re-raises — and whether a vote happens
        # at all is decided from facts every rank shares (nranks, the dp_overlap option), not from this rank's communicator state.
        req = self._dp_overlap_req()                                  # (-2: auto also for a 1-rank communicator, tests)
        will_vote = vote is not None and ((nranks >= 2 and req == -1) or req == -2)
        try:
            _lib.check(self._lib.sdqn_dp_init(self._h, path, unique_id, rank, nranks))
            form = self.dp_form()
            probing = form["form"] == "serial" and form["second_communicator"] and not form["overlap_forced"]
            ok = C.c_int(0)
            if probing:
                _lib.check(self._lib.sdqn_dp_probe(self._h, int(probe_timeout_ms), int(bool(inject_probe_timeout)), C.byref(ok)))
        except Exception:
            if will_vote:
                vote(False)
            raise
        if probing or will_vote:
            agreed = bool(vote(bool(ok.value) and probing)) if will_vote else False
            if probing:
                _lib.check(self._lib.sdqn_dp_set_overlap(self._h, int(agreed)))
            self._dp_vote = dict(local_probe_ok=bool(ok.value), agreed=agreed, voted=will_vote)
        return self.dp_form()

    def _dp_overlap_req(self):
        return getattr(self, "_dp_overlap_opt", -1)

    STEP_STRUCTURES = ("fused", "h16_block_tile", "dp_overlap", "unfused", "generic")
    UPDATE_FORMS = ("single", "dp_serial", "dp_overlap", "grad_only")

    def step_structure(self):
        """(launch structure of a train step, form of its optimizer pass) — DESIGN.md 12's table, as the library itself sees this handle."""
        a, b = C.c_int(), C.c_int()
        _lib.check(self._lib.sdqn_net_step_structure(self._h, C.byref(a), C.byref(b)))
        return self.STEP_STRUCTURES[a.value], self.UPDATE_FORMS[b.value]

    def tuple_counters(self):
        """How train(minibatch) calls have been served: (calls, states read from the device minibatch in place, nothing uploaded at all)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self._lib.sdqn_net_tuple_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def dp_form(self):
        """Which data-parallel form runs: 'none' / 'serial' / 'overlapped', the start-up probe's local result and the vote."""
        f, p, c2 = C.c_int(0), C.c_int(-1), C.c_int(0)
        _lib.check(self._lib.sdqn_dp_form(self._h, C.byref(f), C.byref(p), C.byref(c2)))
        d = dict(form=("none", "serial", "overlapped")[f.value], probe={-1: None, 0: False, 1: True}[p.value],
                 second_communicator=bool(c2.value), overlap_forced=(f.value == 2 and p.value == -1))
        d.update(getattr(self, "_dp_vote", {}))
        return d

    def overflow_steps(self):
        """float16 mode under data parallel: steps skipped because the all-reduced half gradient overflowed."""
        n = C.c_int64()
        _lib.check(self._lib.sdqn_net_overflow_steps(self._h, C.byref(n)))
        return n.value

    def dp_info(self):
        """What RCCL reports about the communicator + the bound device (all -1 without a communicator)."""
        v = [C.c_int(-1) for _ in range(4)]
        _lib.check(self._lib.sdqn_dp_info(self._h, *[C.byref(x) for x in v]))
        return dict(comm_ranks=v[0].value, comm_rank=v[1].value, comm_device=v[2].value, bound_device=v[3].value)

    def apply_update(self, bsz):
        """Optimizer step on the gradient sums currently in the flat buffer (which=3) with divisor bsz: the second half of
        a data-parallel step (see option 'grad_only')."""
        _lib.check(self._lib.sdqn_net_apply_update(self._h, float(bsz)))

    def flat_size(self):
        """values in the flat weight / gradient buffer (float16 networks have no BatchNorm block)"""
        n, tot = C.c_int64(), 0
        for layer in range(5):
            _lib.check(self._lib.sdqn_net_layer_size(self._h, layer, C.byref(n)))
            tot += n.value
        return tot

    def grad_to_half(self):
        """float16 data parallel, first half of the exchange: the flat gradient sums scaled by the payload scale as IEEE half
        (np.float16 array, internal layout) — what a rank hands to the all-reduce."""
        out = np.empty(self.flat_size(), dtype=np.float16)
        _lib.check(self._lib.sdqn_net_grad_to_half(self._h, out.ctypes.data_as(C.POINTER(C.c_uint16)), out.size))
        return out

    def grad_from_half(self, payload):
        """... second half: the SUMMED half payload back into the fp32 gradient buffer; apply_update() then honours its overflow flag."""
        p = np.ascontiguousarray(payload, dtype=np.float16)
        _lib.check(self._lib.sdqn_net_grad_from_half(self._h, p.ctypes.data_as(C.POINTER(C.c_uint16)), p.size))

    def half_payload_state(self):
        v = [C.c_int() for _ in range(3)]
        _lib.check(self._lib.sdqn_net_half_payload_state(self._h, *[C.byref(x) for x in v]))
        return dict(overflow=v[0].value, scale_log2=v[1].value, clean_steps=v[2].value)

    def dp_shutdown(self):
        _lib.check(self._lib.sdqn_dp_shutdown(self._h))


def dp_unique_id(rccl=None):
    buf = C.create_string_buffer(128)
    path = (rccl or _lib.rccl_path()).encode()
    _lib.check(_lib.load().sdqn_dp_unique_id(path, buf))
    return buf.raw
