"""CPU oracle for the replay half of the hot path.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/src/replay_memory.py (ReplayMemory.__init__ :7-24,
add :26-34, getState :37-48, getMinibatch :50-79) in numpy, plus a pure-Python
restatement of CPython 3.10's `random.randint` on top of MT19937 so that the
sampler can be checked without touching the interpreter's global stream.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (simple_dqn_amd/) never does.

Parity status: PINNED for this half — tests/golden/replay_kat.json was produced
by executing the reference's own replay_memory.py under this container's
Python 3.10 (tests/golden/make_replay_golden.py), and tests/test_oracle_replay.py
checks this restatement against those vectors (and against the live reference
module when /root/reference is present).
"""
import random as _pyrandom

import numpy as np


class MT19937:
    """CPython's _random.Random restated (Modules/_randommodule.c): genrand_uint32,
    init_by_array seeding for ints, getrandbits(k<=32), and Lib/random.py's
    _randbelow_with_getrandbits / randint (reference call site replay_memory.py:59)."""

    N, M = 624, 397

    def __init__(self, seed=None):
        self.mt = [0] * self.N
        self.pos = self.N
        if seed is not None:
            self.seed(seed)

    # -- seeding (random.seed(int) -> init_by_array over 32-bit limbs of abs(seed))
    def _init_genrand(self, s):
        mt = self.mt
        mt[0] = s & 0xFFFFFFFF
        for i in range(1, self.N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.pos = self.N

    def seed(self, seed):
        seed = abs(int(seed))
        key = []
        while True:
            key.append(seed & 0xFFFFFFFF)
            seed >>= 32
            if seed == 0:
                break
        self._init_genrand(19650218)
        mt, N = self.mt, self.N
        i, j = 1, 0
        for _ in range(max(N, len(key))):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & 0xFFFFFFFF
            i += 1
            j += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
            if j >= len(key):
                j = 0
        for _ in range(N - 1):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & 0xFFFFFFFF
            i += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
        mt[0] = 0x80000000

    # -- state exchange with `random.getstate()[1]` (624 words + position)
    def setstate(self, state625):
        self.mt = [int(x) & 0xFFFFFFFF for x in state625[:624]]
        self.pos = int(state625[624])

    def getstate(self):
        return tuple(self.mt) + (self.pos,)

    def genrand_uint32(self):
        mt, N, M = self.mt, self.N, self.M
        if self.pos >= N:
            for kk in range(N):
                y = (mt[kk] & 0x80000000) | (mt[(kk + 1) % N] & 0x7FFFFFFF)
                mt[kk] = mt[(kk + M) % N] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.pos = 0
        y = mt[self.pos]
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def getrandbits(self, k):
        assert 0 < k <= 32
        return self.genrand_uint32() >> (32 - k)

    def randbelow(self, n):
        k = n.bit_length()
        r = self.getrandbits(k)
        while r >= n:
            r = self.getrandbits(k)
        return r

    def randint(self, a, b):
        return a + self.randbelow(b - a + 1)


class ReplayOracle:
    """numpy restatement of the reference ReplayMemory (same attribute names)."""

    def __init__(self, size, screen_height=84, screen_width=84, history_length=4, batch_size=32):
        self.size = size
        self.actions = np.empty(size, dtype=np.uint8)            # replay_memory.py:10
        self.rewards = np.empty(size, dtype=np.int64)            # :11 (np.integer -> int64 on Linux)
        self.screens = np.empty((size, screen_height, screen_width), dtype=np.uint8)  # :12
        self.terminals = np.empty(size, dtype=np.bool_)          # :13
        self.history_length = history_length
        self.dims = (screen_height, screen_width)
        self.batch_size = batch_size
        self.count = 0
        self.current = 0
        self.prestates = np.empty((batch_size, history_length) + self.dims, dtype=np.uint8)   # :21
        self.poststates = np.empty((batch_size, history_length) + self.dims, dtype=np.uint8)  # :22

    def add(self, action, reward, screen, terminal):             # :26-34
        assert screen.shape == self.dims
        self.actions[self.current] = action
        self.rewards[self.current] = reward
        self.screens[self.current, ...] = screen
        self.terminals[self.current] = terminal
        self.count = max(self.count, self.current + 1)
        self.current = (self.current + 1) % self.size

    def getState(self, index):                                   # :37-48
        assert self.count > 0
        index = index % self.count
        h = self.history_length
        if index >= h - 1:
            return self.screens[(index - (h - 1)):(index + 1), ...]
        indexes = [(index - i) % self.count for i in reversed(range(h))]
        return self.screens[indexes, ...]

    def sample_indexes(self, rng=None):                          # :54-68
        """rng: an MT19937 (above) or None for the interpreter's global `random`."""
        randint = _pyrandom.randint if rng is None else rng.randint
        assert self.count > self.history_length
        h = self.history_length
        indexes = []
        while len(indexes) < self.batch_size:
            while True:
                index = randint(h, self.count - 1)
                if index >= self.current and index - h < self.current:
                    continue
                if self.terminals[(index - h):index].any():
                    continue
                break
            indexes.append(index)
        return indexes

    def gather(self, indexes):                                   # :71-79
        for k, index in enumerate(indexes):
            self.prestates[k, ...] = self.getState(index - 1)
            self.poststates[k, ...] = self.getState(index)
        actions = self.actions[indexes]
        rewards = self.rewards[indexes]
        terminals = self.terminals[indexes]
        return self.prestates, actions, rewards, self.poststates, terminals

    def getMinibatch(self, rng=None):                            # :50-79
        return self.gather(self.sample_indexes(rng))


def synthetic_fill(mem, seed, num_actions=4, count=None, current=None):
    """The §8c/§8d synthetic ring fill: works on the oracle, the
    reference module and the product ReplayMemory alike (anything exposing the
    reference's numpy attributes)."""
    size = mem.size
    rng = np.random.RandomState(seed)
    mem.screens[:] = rng.randint(0, 256, size=mem.screens.shape, dtype=np.uint8)
    mem.actions[:] = rng.randint(0, num_actions, size=size).astype(np.uint8)
    mem.rewards[:] = rng.randint(-1, 2, size=size)
    mem.terminals[:] = rng.rand(size) < 0.005
    mem.count = size if count is None else count
    mem.current = (size // 3) if current is None else current
    return mem
