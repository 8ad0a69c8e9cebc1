"""CPU test of the lazy minibatch views (simple_dqn_amd/_lazy.py): what getMinibatch() hands out for prestates / poststates since
round 5 (reference: replay_memory.py:21-22,76-79 returns the preallocated arrays themselves)."""
import copy
import pickle

import numpy as np
import pytest

from simple_dqn_amd._lazy import LazyMinibatchArray


class _Mem:
    batch_size, history_length, dims = 4, 2, (3, 5)

    def __init__(self):
        self.fetches = 0
        self.buf = {"pre": np.arange(120, dtype=np.uint8).reshape(4, 2, 3, 5), "post": np.zeros((4, 2, 3, 5), np.uint8)}

    def _states(self, which):
        self.fetches += 1
        return self.buf[which]


def test_metadata_needs_no_fetch_and_every_look_at_the_data_does():
    m = _Mem()
    x = LazyMinibatchArray(m, "pre")
    assert x.shape == (4, 2, 3, 5) and len(x.shape) == 4 and x.dtype == np.uint8 and len(x) == 4 and x.ndim == 4 and x.size == 120
    assert m.fetches == 0
    assert np.asarray(x) is m.buf["pre"] and m.fetches == 1                      # the aliased buffer itself, not a copy
    assert x[1, 0, 2, 3] == m.buf["pre"][1, 0, 2, 3] and m.fetches == 2
    assert np.array_equal(x, m.buf["pre"]) and (x == m.buf["pre"]).all() and int(np.sum(x)) == int(m.buf["pre"].sum())
    assert x.copy().base is None and x.astype(np.float32).dtype == np.float32 and x.mean() == m.buf["pre"].mean()
    assert np.ascontiguousarray(x, dtype=np.uint8) is m.buf["pre"] or np.shares_memory(np.ascontiguousarray(x, dtype=np.uint8), m.buf["pre"])
    assert [r.shape for r in x] == [(2, 3, 5)] * 4
    assert ((x + 1) == (m.buf["pre"] + 1)).all() and ((1 + x) == (m.buf["pre"] + 1)).all()
    assert np.transpose(x, (1, 2, 3, 0)).shape == (2, 3, 5, 4)                    # deepqnetwork.py:96's use of the states
    before = m.fetches
    x[0] = 9                                                                      # writes go to the aliased buffer (and fetch first)
    assert m.fetches == before + 1 and (m.buf["pre"][0] == 9).all()
    with pytest.raises(TypeError):
        memoryview(x)                                                             # no door to the pinned bytes that skips the fetch
    y = pickle.loads(pickle.dumps(x))
    assert isinstance(y, np.ndarray) and np.array_equal(y, m.buf["pre"]) and not np.shares_memory(y, m.buf["pre"])
    assert isinstance(copy.deepcopy(x), np.ndarray)
    with pytest.raises(TypeError):
        hash(x)
