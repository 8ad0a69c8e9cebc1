"""Generates tests/golden/replay_kat.json by EXECUTING the reference's own
replay_memory.py (imported unmodified from /root/reference/src) under this
interpreter.  Run only where /root/reference exists; the JSON it writes is the
committed fixture the oracle and the HIP path are pinned against.
"""
import json, os, random, sys, warnings, zlib
import numpy as np

warnings.simplefilter("ignore")
sys.path.insert(0, "/root/reference/src")
from replay_memory import ReplayMemory  # noqa: E402  (the reference module itself)


class Args:
    screen_height = 84
    screen_width = 84
    history_length = 4
    batch_size = 32


def crc(a):
    return "%08x" % (zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)


def kat(size, B, current, count, seed, ncalls=1):
    a = Args()
    a.batch_size = B
    m = ReplayMemory(size, a)
    rng = np.random.RandomState(0)
    m.screens[:] = rng.randint(0, 256, size=m.screens.shape, dtype=np.uint8)
    m.actions[:] = rng.randint(0, 4, size=size).astype(np.uint8)
    m.rewards[:] = rng.randint(-1, 2, size=size)
    m.terminals[:] = rng.rand(size) < 0.005
    m.count, m.current = count, current
    random.seed(seed)
    # record accepted indexes by wrapping randint draws
    draws = []
    orig = random.randint
    def spy(a_, b_):
        v = orig(a_, b_); draws.append(v); return v
    random.randint = spy
    calls = []
    try:
        for _ in range(ncalls):
            n0 = len(draws)
            pre, act, rew, post, term = m.getMinibatch()
            # accepted indexes: recover by matching actions is ambiguous; recompute via the rule
            acc = []
            for idx in draws[n0:]:
                if idx >= m.current and idx - 4 < m.current:
                    continue
                if m.terminals[idx - 4:idx].any():
                    continue
                acc.append(idx)
            assert len(acc) == B
            assert (act == m.actions[acc]).all()
            calls.append(dict(draws=len(draws) - n0, indexes=[int(i) for i in acc],
                              crc_pre=crc(pre), crc_post=crc(post), crc_actions=crc(act),
                              crc_rewards=crc(rew.astype(np.int64)), crc_terminals=crc(term),
                              rewards_dtype=str(rew.dtype), terminals_dtype=str(term.dtype)))
    finally:
        random.randint = orig
    mt_after = [int(x) for x in random.getstate()[1]]
    return dict(size=size, B=B, current=current, count=count, seed=seed, fill_seed=0,
                calls=calls, mt_after_crc=crc(np.array(mt_after, dtype=np.uint32)))


out = dict(
    python=sys.version.split()[0], numpy=np.__version__,
    mt_kat=dict(seed=42, a=4, b=9999, values=(random.seed(42), [random.randint(4, 9999) for _ in range(5)])[1]),
    kats=[kat(10000, 32, 123, 10000, 1234), kat(10000, 32, 5000, 5000, 7), kat(10000, 256, 9998, 10000, 99),
          kat(10000, 32, 3333, 10000, 2024, ncalls=3), kat(600, 32, 10, 600, 5)],
)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "replay_kat.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path)
for k in out["kats"]:
    print(k["size"], k["B"], k["seed"], [c["draws"] for c in k["calls"]], k["calls"][0]["indexes"][:6], k["calls"][0]["crc_pre"])
