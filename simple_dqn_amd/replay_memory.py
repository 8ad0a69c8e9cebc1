"""ReplayMemory — drop-in for /root/reference/src/replay_memory.py:6-79.

Same constructor, methods, attributes and return types as the reference class; the ring's master
copy lives in pinned host DRAM (the numpy attributes are views of it) with a mirror in HBM, and
getMinibatch() gathers (s, a, r, s', terminal) with a HIP kernel (sdqn_replay_gather).  Index
sampling consumes Python's global `random` stream exactly like the reference (:59), natively.
"""
import array
import ctypes as C
import logging
import random

import numpy as np

from . import _lib
from ._lazy import LazyMinibatchArray
from ._tracked import DirtySlots, TrackedArray

logger = logging.getLogger(__name__)
HBM_MIRROR, ZERO_COPY = 1, 2


class ReplayMemory:
    def __init__(self, size, args, flags=HBM_MIRROR):
        self._lib = _lib.load()
        _lib.bind_device(args)              # main.py builds the memory first: --device_id must be bound before its allocations
        self.size = size
        self.history_length = args.history_length
        self.dims = (args.screen_height, args.screen_width)
        self.batch_size = args.batch_size
        h = C.c_void_p()
        _lib.check(self._lib.sdqn_replay_create(C.byref(h), size, self.dims[0], self.dims[1],
                                                self.history_length, self.batch_size, flags))
        self._h = h
        ps, pa, pr, pt = _lib._u8p(), _lib._u8p(), _lib._i64p(), _lib._u8p()
        _lib.check(self._lib.sdqn_replay_host_ptrs(h, C.byref(ps), C.byref(pa), C.byref(pr), C.byref(pt)))
        # replay_memory.py:10-13 — same dtypes (np.integer resolves to int64 on Linux).  The reference has ONE copy of the ring;
        # here these are views of the pinned master copy and the kernels read an HBM mirror, so the views TRACK in-place writes
        # (_tracked.py) and the slots they touched are uploaded before the next device use (_check_mirror): coherent like one copy
        self._flags = flags
        self._dirty_frames, self._dirty_meta = DirtySlots(), DirtySlots()
        raw = {"actions": np.ctypeslib.as_array(pa, shape=(size,)), "rewards": np.ctypeslib.as_array(pr, shape=(size,)),
               "screens": np.ctypeslib.as_array(ps, shape=(size,) + self.dims),
               "terminals": np.ctypeslib.as_array(pt, shape=(size,)).view(np.bool_)}
        self._ring_base = {k: (v.__array_interface__["data"][0], v.nbytes // size) for k, v in raw.items()}   # (address, bytes per slot)
        # the public attributes are READ-ONLY for numpy (an alias that escapes the tracking cannot be written); the tracked write
        # paths go through these private writable byte views of the same memory (_tracked.py: TrackedArray._w)
        self._raw = raw
        self._raw_bytes = {k: v.reshape(-1).view(np.uint8) for k, v in raw.items()}
        self.actions, self.rewards = TrackedArray(raw["actions"], self, "actions"), TrackedArray(raw["rewards"], self, "rewards")
        self.screens, self.terminals = TrackedArray(raw["screens"], self, "screens"), TrackedArray(raw["terminals"], self, "terminals")
        mp, mq, ma, mr, mt = _lib._u8p(), _lib._u8p(), _lib._u8p(), _lib._i64p(), _lib._u8p()
        _lib.check(self._lib.sdqn_replay_minibatch_ptrs(h, C.byref(mp), C.byref(mq), C.byref(ma), C.byref(mr), C.byref(mt)))
        shp = (self.batch_size, self.history_length) + self.dims
        # :21-22, reused every call (aliased, like the reference).  Tracked too: while nobody has written into them since the last
        # getMinibatch(), DeepQNetwork.train(minibatch) lets the step read the device copy of the gathered states in place instead of
        # uploading them again (sdqn_replay_declare_minibatch_clean)
        self._mb_dirty = True
        self._mb_pending = False                # a gather has run whose states have not been copied into the host buffers yet (_lazy.py)
        raw_mb = {"mb_pre": np.ctypeslib.as_array(mp, shape=shp), "mb_post": np.ctypeslib.as_array(mq, shape=shp)}
        self._raw_bytes.update({k: v.reshape(-1).view(np.uint8) for k, v in raw_mb.items()})
        self._raw_mb = raw_mb
        self._mb_ptrs = (_lib.ptr(raw_mb["mb_pre"], C.c_uint8), _lib.ptr(raw_mb["mb_post"], C.c_uint8))    # (constant: the handle's pinned buffers)
        self._prestates = TrackedArray(raw_mb["mb_pre"], self, "mb_pre")
        self._poststates = TrackedArray(raw_mb["mb_post"], self, "mb_post")
        self._lazy_pre, self._lazy_post = LazyMinibatchArray(self, "pre"), LazyMinibatchArray(self, "post")
        self._mb_actions = np.ctypeslib.as_array(ma, shape=(self.batch_size,))
        self._mb_rewards = np.ctypeslib.as_array(mr, shape=(self.batch_size,))
        self._mb_terminals = np.ctypeslib.as_array(mt, shape=(self.batch_size,)).view(np.bool_)
        self._idx = np.empty(self.batch_size, dtype=np.int64)
        self._mt = (C.c_uint32 * _lib.MT_WORDS)()
        self.last_indexes = None
        logger.info("Replay memory size: %d" % self.size)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and self._lib is not None:
            self._lib.sdqn_replay_destroy(h)

    # replay_memory.py:21-22: the preallocated minibatch buffers as attributes.  Reading them fetches the last gather's states first
    # (getMinibatch() itself no longer waits for that copy: _lazy.py); the same arrays every time, aliased like the reference's
    # (the attributes ARE what getMinibatch() returns, like the reference's `return self.prestates, ..., self.poststates, ...`:
    #  `pre is mem.prestates` holds; np.asarray(mem.prestates) is the pinned buffer itself)
    @property
    def prestates(self):
        return self._lazy_pre

    @property
    def poststates(self):
        return self._lazy_post

    def _states(self, which):
        self._materialize()
        return self._prestates if which == "pre" else self._poststates

    def _materialize(self):
        if self._mb_pending:
            _lib.check(self._lib.sdqn_replay_minibatch_to_host(self._h))     # D2H of [pre | post] (+ the three small arrays) and the wait
            self._mb_pending = False
            self._mb_dirty = False              # host and device copies of the minibatch are identical from here on

    def _device_minibatch_gen(self):
        g = C.c_uint64()
        _lib.check(self._lib.sdqn_replay_minibatch_gen(self._h, C.byref(g), None))
        return g.value

    # count / current live in the native handle (add() updates them there)
    def _state(self):
        c, k = C.c_int64(), C.c_int64()
        _lib.check(self._lib.sdqn_replay_get_state(self._h, C.byref(c), C.byref(k)))
        return c.value, k.value

    @property
    def count(self):
        return self._state()[0]

    # Backstop behind the write tracking: slots that a direct assignment of count / current newly EXPOSES to the sampler
    # are uploaded before the next device use whether or not a tracked write touched them — a bulk fill through raw pointers followed
    # by `mem.count = n` trains on what the host holds.  Costs the newly exposed slots only.
    def _expose(self, first, last):
        if last > first:
            if self._flags != ZERO_COPY:
                self._dirty_frames.mark(first, last)
            self._dirty_meta.mark(first, last)

    @count.setter
    def count(self, v):
        old, cur = self._state()
        _lib.check(self._lib.sdqn_replay_set_state(self._h, int(v), cur))
        self._expose(old, min(int(v), self.size))

    @property
    def current(self):
        return self._state()[1]

    @current.setter
    def current(self, v):
        cnt, old = self._state()
        _lib.check(self._lib.sdqn_replay_set_state(self._h, cnt, int(v)))
        v = int(v)
        if v >= old:
            self._expose(old, v)
        else:                                           # moved across the wrap
            self._expose(old, self.size); self._expose(0, v)

    # ---- coherence of the HBM mirror with the numpy views -----------------------------------------------------------
    def _mark_dirty_bytes(self, kind, lo, hi):
        """Called by the tracked views: memory [lo, hi) of ring array `kind` was written in place."""
        if kind.startswith("mb_"):                      # the minibatch buffers: the host copy no longer equals the device copy
            self._mb_dirty = True
            return
        if self._flags == ZERO_COPY and kind == "screens":
            return                                      # the kernels read the pinned frames themselves ...
        # ... but NOT the raw actions / rewards / terminals: also a zero-copy ring's kernels read the PACKED MetaRec array, which only
        # sdqn_replay_upload_meta re-packs from these views (a rewards edit trained on stale metadata there)
        base, bps = self._ring_base[kind]
        first, last = max(0, (lo - base) // bps), min(self.size, -((base - hi) // bps))     # floor / ceil in slots
        if last > first:
            (self._dirty_frames if kind == "screens" else self._dirty_meta).mark(first, last)

    def _check_mirror(self):
        """Before every device use of the ring: slots written through the numpy views since the last upload go to the HBM
        mirror now (frames and packed metadata separately — a rewards-only edit does not re-send 7 KB per slot)."""
        if not self._dirty_frames.iv and not self._dirty_meta.iv:         # (the common case: nothing was written through the views)
            return
        frames = self._dirty_frames.take()
        for lo, hi in frames:
            _lib.check(self._lib.sdqn_replay_upload(self._h, lo, hi - lo))           # (also re-packs + sends the range's metadata)
        for lo, hi in self._dirty_meta.take():
            if not any(a <= lo and hi <= b for a, b in frames):
                _lib.check(self._lib.sdqn_replay_upload_meta(self._h, lo, hi - lo))

    @property
    def mirror_dirty(self):
        """(frames, metadata) slot ranges written through the views and not yet uploaded — None when clean."""
        return (list(self._dirty_frames.iv) or None, list(self._dirty_meta.iv) or None)

    def add(self, action, reward, screen, terminal):               # :26-34
        assert screen.shape == self.dims
        scr = np.ascontiguousarray(screen, dtype=np.uint8)
        _lib.check(self._lib.sdqn_replay_add(self._h, int(action), int(reward), _lib.ptr(scr, C.c_uint8), int(bool(terminal))))

    def getState(self, index):                                     # :37-48 (host views; not on the train path)
        count = self.count
        assert count > 0, "replay memory is empy, use at least --random_steps 1"
        index = index % count
        if index >= self.history_length - 1:
            return self.screens[(index - (self.history_length - 1)):(index + 1), ...]
        indexes = [(index - i) % count for i in reversed(range(self.history_length))]
        return self.screens[indexes, ...]

    def sync_mirror(self, first=0, n=None):
        """Explicit upload of ring slots [first, first + n) into the HBM mirror.  Not needed after writes through the numpy
        attributes (tracked, uploaded automatically before the next device use); needed after writes that bypass numpy
        (buffer protocol: memoryview / readinto / ctypes)."""
        n = self.size - first if n is None else n
        _lib.check(self._lib.sdqn_replay_upload(self._h, first, n))
        if first == 0 and n == self.size:
            self._dirty_frames.take(); self._dirty_meta.take()

    def sample_indexes(self):
        """replay_memory.py:54-68 on Python's GLOBAL random stream (shared with agent.py:32,50-51): the generator's state goes in as a
        copy, the library's sampler draws from it, and Python's own generator is advanced by exactly the 32-bit words that were drawn
        (one getrandbits call: Modules/_randommodule.c — rebuilding a 625-tuple for random.setstate cost 40 us per call)."""
        st = random.getstate()
        arr = array.array("I", st[1])
        mt = (C.c_uint32 * _lib.MT_WORDS).from_buffer(arr)
        w0, w1 = C.c_uint64(), C.c_uint64()
        self._lib.sdqn_mt_words(C.byref(w0))
        try:
            _lib.check(self._lib.sdqn_replay_sample(self._h, mt, _lib.ptr(self._idx, C.c_int64), None))
        finally:
            self._lib.sdqn_mt_words(C.byref(w1))
            if w1.value != w0.value:
                random.getrandbits(32 * (w1.value - w0.value))
        return self._idx

    def gather(self, indexes):
        """replay_memory.py:71-79 by index: the HIP gather is ENQUEUED (states into the device minibatch) and the call returns at once.
        prestates / poststates come back as lazy views of the aliased buffers (their host copy is fetched on first access, _lazy.py;
        DeepQNetwork.train consumes the device copy directly while nobody has looked); actions / rewards / terminals are
        `ring[indexes]` of the host master copy — fresh arrays, exactly the reference's fancy-index copies (:76-78)."""
        idx = np.ascontiguousarray(indexes, dtype=np.int64)
        assert idx.shape == (self.batch_size,)
        self._check_mirror()
        _lib.check(self._lib.sdqn_replay_gather(self._h, _lib.ptr(idx, C.c_int64)))
        self._mb_pending = True
        self._mb_dirty = False
        if self._flags == ZERO_COPY:
            # the enqueued gather reads the pinned ring ITSELF (no mirror): a host store by the next add() or a tracked write could race
            # with the still-queued kernel (the sampler lets slot `current` fall inside a sampled window) — this flag keeps the
            # synchronous fetch of rounds 1-4
            self._materialize()
        self.last_indexes = idx.copy()
        raw = self._raw
        return self._lazy_pre, raw["actions"][idx], raw["rewards"][idx], self._lazy_post, raw["terminals"][idx]

    def getMinibatch(self):                                        # :50-79
        assert self.count > self.history_length
        return self.gather(self.sample_indexes())

    # ---- checkpoint of the ring (additive: the reference never persists its replay memory, README.md:132) -----------
    _MAGIC = b"SDQNRING1\n"

    def save(self, path):
        """Raw dump of the filled part of the ring (header + actions, rewards, terminals, screens), streamed straight
        from the pinned master copy: 7 GB at 1 M frames, no extra host copy."""
        count, current = self._state()
        with open(path, "wb") as f:
            f.write(self._MAGIC)
            np.array([self.size, count, current, self.dims[0], self.dims[1], self.history_length], dtype=np.int64).tofile(f)
            self.actions[:count].tofile(f); self.rewards[:count].tofile(f)
            self.terminals[:count].view(np.uint8).tofile(f); self.screens[:count].tofile(f)

    def load(self, path):
        """Restores a ring written by save() into THIS memory (same size and geometry) and refreshes the HBM mirror."""
        with open(path, "rb") as f:
            assert f.read(len(self._MAGIC)) == self._MAGIC, "not a replay-memory checkpoint"
            size, count, current, h, w, hist = np.fromfile(f, dtype=np.int64, count=6)
            assert (size, h, w, hist) == (self.size, self.dims[0], self.dims[1], self.history_length), "geometry mismatch"
            self._raw["actions"][:count] = np.fromfile(f, dtype=np.uint8, count=count)
            self._raw["rewards"][:count] = np.fromfile(f, dtype=np.int64, count=count)
            self._raw["terminals"][:count] = np.fromfile(f, dtype=np.uint8, count=count).view(np.bool_)
            f.readinto(memoryview(self._raw["screens"][:count]).cast("B"))
        _lib.check(self._lib.sdqn_replay_set_state(self._h, int(count), int(current)))
        self.sync_mirror(0, int(count))

    def bench_gather(self, indexes, iters=100):
        """ms per launch of the standalone gather kernel.  `indexes`: one index set [B] (repeated: cache-resident after the first
        launch) or several [nsets, B] (cycled, one set per launch — what consecutive getMinibatch() calls look like to the memory system)."""
        idx = np.ascontiguousarray(indexes, dtype=np.int64)
        ms = C.c_float()
        self._materialize()                     # (the timed launches overwrite the device minibatch: a pending getMinibatch() is fetched first)
        if idx.ndim == 2:
            assert idx.shape[1] == self.batch_size
            _lib.check(self._lib.sdqn_replay_bench_gather_sets(self._h, _lib.ptr(idx, C.c_int64), idx.shape[0], iters, C.byref(ms)))
        else:
            _lib.check(self._lib.sdqn_replay_bench_gather(self._h, _lib.ptr(idx, C.c_int64), iters, C.byref(ms)))
        return ms.value
