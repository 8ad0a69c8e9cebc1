"""Write tracking of the replay ring's numpy views (simple_dqn_amd/_tracked.py) — host logic, no GPU:
every in-place write path must report the ring slots it touched, reads and fresh results must not."""
import numpy as np

from simple_dqn_amd._tracked import DirtySlots, TrackedArray


class Owner:
    def __init__(self, size=100, row=(6, 5)):
        self.raw = np.zeros((size,) + row, np.uint8)
        self.base, self.bps, self.size = self.raw.__array_interface__["data"][0], self.raw.nbytes // size, size
        self.dirty = DirtySlots()
        self._raw_bytes = {"screens": self.raw.reshape(-1).view(np.uint8)}       # the private writable alias (ReplayMemory keeps the same)
        self.a = TrackedArray(self.raw, self, "screens")

    def _mark_dirty_bytes(self, kind, lo, hi):
        assert kind == "screens"
        first, last = max(0, (lo - self.base) // self.bps), min(self.size, -((self.base - hi) // self.bps))
        if last > first:
            self.dirty.mark(first, last)

    def take(self):
        return self.dirty.take()


def test_every_write_path_reports_its_slots():
    o = Owner()
    a = o.a
    a[3] = 7;                         assert o.take() == [(3, 4)]
    a[10:20] = 1;                     assert o.take() == [(10, 20)]
    a[5, 2, 1] = 9;                   assert o.take() == [(5, 6)]
    a[30:40][3] = 2;                  assert o.take() == [(33, 34)]                 # view of a view
    a[-1] = 4;                        assert o.take() == [(99, 100)]
    a[[1, 50, 70]] = 3;               assert o.take() == [(1, 2), (50, 51), (70, 71)]  # fancy index on the first axis: the rows it names
    a[np.array([7, 3])] = 4;          assert o.take() == [(3, 4), (7, 8)]
    a[a[:, 0, 0] == 3] = 0;           assert o.take() == [(1, 2), (50, 51), (70, 71)]  # boolean mask on the first axis
    a[list(range(10, 90))] = 6;       assert o.take() == [(10, 90)]                 # many rows: their span
    a[[2, 5], 1, :] = 1;              assert o.take() == [(2, 3), (5, 6)]
    a[...] = 5;                       assert o.take() == [(0, 100)]
    v = a[40:60]
    v += 1;                           assert o.take() == [(40, 60)]                 # in-place operator
    np.bitwise_xor(a[60:64], np.uint8(3), out=a[60:64])
    assert o.take() == [(60, 64)]                                                   # ufunc out= (bench.fill_ring's form)
    np.copyto(a[70:72], 8);           assert o.take() == [(70, 72)]
    a[80:85].fill(1);                 assert o.take() == [(80, 85)]
    np.add.at(a, (slice(2, 4),), 1);  assert o.take() == [(0, 100)]
    a.reshape(100, 30)[7, 3] = 1;     assert o.take() == [(7, 8)]                   # reshaped view of the same memory
    a.view(np.int8)[8] = -1;          assert o.take() == [(8, 9)]
    np.put(a[90:92], [0, 1], 5);      assert o.take() == [(90, 92)]
    assert (o.raw[80:85] == 1).all() and (o.raw[70:72] == 8).all()                  # (the writes themselves happened)


def test_reads_and_fresh_results_do_not_mark():
    o = Owner()
    a = o.a
    _ = a[3].sum(); _ = a[10:20].copy(); b = a + 1; c = np.asarray(a)[5]; d = a.astype(np.float32) / 255
    assert not isinstance(b, TrackedArray) or b.base is None
    b[0] = 9                                                                        # a fresh array aliases nothing
    _ = np.concatenate([a[:2], a[5:6]]); _ = a.mean(); _ = a[::7].max()
    assert o.take() == []


def test_scalar_ring_arrays_and_bool_view():
    class O1(Owner):
        def __init__(self):
            self.raw = np.zeros(50, np.int64); self.size = 50
            self.base, self.bps = self.raw.__array_interface__["data"][0], 8
            self.dirty = DirtySlots(); self._raw_bytes = {"screens": self.raw.view(np.uint8)}; self.a = TrackedArray(self.raw, self, "screens")
    o = O1()
    o.a[7] = -1;                      assert o.take() == [(7, 8)]
    o.a[10:13] = [1, 2, 3];           assert o.take() == [(10, 13)]
    o.a[:] = np.arange(50);           assert o.take() == [(0, 50)]
    t = TrackedArray(np.zeros(20, np.uint8).view(np.bool_), o, "screens")           # other memory, outside the ring: no writable alias -> loud
    import pytest
    with pytest.raises(ValueError):
        t[3] = True
    assert o.take() == []


def test_dirty_slots_merge_and_cap():
    d = DirtySlots()
    d.mark(10, 20); d.mark(30, 40); d.mark(20, 25); d.mark(0, 1)
    assert d.iv == [(0, 1), (10, 25), (30, 40)] and d.lo == 0 and d.hi == 40
    d.mark(24, 31)
    assert d.take() == [(0, 1), (10, 40)] and not d
    for i in range(100):
        d.mark(10 * i, 10 * i + 1)
    assert len(d.iv) <= DirtySlots.MAX and d.lo == 0 and d.hi == 991
    assert all(any(a <= 10 * i and 10 * i + 1 <= b for a, b in d.iv) for i in range(100))      # nothing lost


def test_untracked_aliases_are_read_only():
    """an alias that escapes the tracking (np.asarray, .view(np.ndarray), memoryview) cannot be written — a write
    raises instead of leaving the HBM mirror stale silently; the tracked object itself writes through its private alias."""
    import pytest
    o = Owner()
    for alias in (np.asarray(o.a), o.a.view(np.ndarray), np.asarray(o.a[10:20]), o.a[5].view(np.ndarray)):
        assert not alias.flags.writeable
        with pytest.raises(ValueError):
            alias[0] = 7
    with pytest.raises((TypeError, ValueError, BufferError)):
        memoryview(o.a)[0] = b"\x01"
    assert o.take() == [] and not o.raw.any()
    o.a[5] = 7                                                                      # the tracked door still works, and reports
    assert o.take() == [(5, 6)] and (o.raw[5] == 7).all()
    c = o.a[[1, 2]]                                                                 # a fancy-index COPY owns its memory: writable, untracked
    c[0] = 9
    assert o.take() == [] and not (o.raw[1] == 9).any()
    d = o.a.copy(); d[3] = 1
    assert o.take() == []
