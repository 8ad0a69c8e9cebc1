"""Write-tracking numpy views of the replay ring.

The reference keeps ONE copy of the ring (replay_memory.py:10-13): whatever a caller writes into `mem.screens`,
`mem.actions`, `mem.rewards` or `mem.terminals` is what the next getMinibatch() reads.  Here those attributes are views of
the pinned host master copy and the kernels read an HBM mirror, so a direct write must reach the mirror before the next
gather.  `TrackedArray` is an ndarray subclass whose every in-place write path records WHICH ring slots it touched in the
owning ReplayMemory (a dirty slot range); the memory uploads exactly that range before its next device use.  Nothing is
ever trained on stale frames silently, and no `sync_mirror()` call is needed for correctness any more (it remains as an
explicit bulk upload).

Write paths covered: indexing assignment (`a[...] = v`, any index kind), in-place operators and every ufunc / numpy
function with `out=` (`np.bitwise_xor(x, y, out=a[s:e])`), `np.copyto / put / place / putmask / put_along_axis`, and the
mutating ndarray methods (`fill`, `put`, `sort`, `partition`, `byteswap(inplace=True)`, `setfield`, `itemset`).  Views of
views stay tracked (`a[10:20][3] = v`), also after `.view(dtype)` / `.reshape` of the same memory.  NOT covered: writes
through the raw buffer protocol (`memoryview(a)`, `file.readinto`, ctypes pointers) — those callers own their
`sync_mirror()` (ReplayMemory.load does).
"""
import weakref

import numpy as np

# numpy functions whose FIRST positional argument (or the named one) is written in place
_MUTATING_FUNCS = {"copyto": "dst", "put": "a", "place": "arr", "putmask": "a", "put_along_axis": "arr", "fill_diagonal": "a"}


class TrackedArray(np.ndarray):
    """ndarray view that reports in-place writes to its owner: owner._mark_dirty_bytes(kind, lo_addr, hi_addr)."""

    def __new__(cls, base_array, owner, kind):
        obj = np.asarray(base_array).view(cls)
        obj._owner = weakref.ref(owner)
        obj._kind = kind
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)
        self._kind = getattr(obj, "_kind", None)

    # ---- reporting ----------------------------------------------------------------------------------------------
    def _touched(self):
        ref = self._owner
        owner = ref() if ref is not None else None
        if owner is None or self.size == 0:
            return
        lo, hi = _byte_bounds(self)
        owner._mark_dirty_bytes(self._kind, lo, hi)

    # ---- write paths --------------------------------------------------------------------------------------------
    def __setitem__(self, key, value):
        # an int / slice on the first axis narrows the range to the slots actually written; anything else marks this view's extent
        target = self
        try:
            if isinstance(key, (int, np.integer, slice)):
                target = np.ndarray.__getitem__(self, key)
            elif isinstance(key, tuple) and key and isinstance(key[0], (int, np.integer, slice)):
                target = np.ndarray.__getitem__(self, key[0])
        except Exception:
            target = self
        np.ndarray.__setitem__(self, key, value)
        if isinstance(target, TrackedArray):
            target._touched()
        else:                                   # a 0-d element (a[i] = v on a 1-d array): one slot
            self._touched_element(key)

    def _touched_element(self, key):
        ref = self._owner
        owner = ref() if ref is not None else None
        if owner is None:
            return
        lo, hi = _byte_bounds(self)
        try:
            k = key[0] if isinstance(key, tuple) else key
            i = int(k)
            if i < 0:
                i += self.shape[0]
            st = self.strides[0]
            if st > 0:
                owner._mark_dirty_bytes(self._kind, lo + i * st, lo + i * st + self.itemsize)
                return
        except Exception:
            pass
        owner._mark_dirty_bytes(self._kind, lo, hi)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        out = kwargs.get("out")
        outs = out if isinstance(out, tuple) else ((out,) if out is not None else ())
        written = [o for o in outs if isinstance(o, TrackedArray)]
        if method == "at" and inputs and isinstance(inputs[0], TrackedArray):      # np.add.at(a, idx, v) writes its first input
            written.append(inputs[0])
        inputs = tuple(x.view(np.ndarray) if isinstance(x, TrackedArray) else x for x in inputs)
        if outs:
            kwargs["out"] = tuple(o.view(np.ndarray) if isinstance(o, TrackedArray) else o for o in outs)
        res = getattr(ufunc, method)(*inputs, **kwargs)
        for w in written:
            w._touched()
        if method == "at":
            return None
        if outs:                                # hand the tracked objects back (a += 1 rebinds the name to the result)
            if isinstance(res, tuple):
                return tuple(o if isinstance(o, TrackedArray) else r for o, r in zip(outs, res))
            return outs[0] if isinstance(outs[0], TrackedArray) else res
        return res                              # a fresh result aliases nothing: plain ndarray

    def __array_function__(self, func, types, args, kwargs):
        written = []
        name = getattr(func, "__name__", "")
        if name in _MUTATING_FUNCS:
            tgt = args[0] if args else kwargs.get(_MUTATING_FUNCS[name])
            if isinstance(tgt, TrackedArray):
                written.append(tgt)
        o = kwargs.get("out")
        for x in (o if isinstance(o, tuple) else (o,)):
            if isinstance(x, TrackedArray):
                written.append(x)
        res = super().__array_function__(func, types, args, kwargs)
        for w in written:
            w._touched()
        return res

    def fill(self, value):
        np.ndarray.fill(self, value); self._touched()

    def put(self, *a, **k):
        np.ndarray.put(self, *a, **k); self._touched()

    def sort(self, *a, **k):
        np.ndarray.sort(self, *a, **k); self._touched()

    def partition(self, *a, **k):
        np.ndarray.partition(self, *a, **k); self._touched()

    def setfield(self, *a, **k):
        np.ndarray.setfield(self, *a, **k); self._touched()

    def byteswap(self, inplace=False):
        r = np.ndarray.byteswap(self, inplace)
        if inplace:
            self._touched()
        return r

    def itemset(self, *a):                      # (numpy < 2)
        np.ndarray.itemset(self, *a); self._touched()


def _byte_bounds(a):
    """[lo, hi) addresses of the memory an array view can touch (numpy's byte_bounds without the deprecation churn)."""
    lo = hi = a.__array_interface__["data"][0]
    for n, st in zip(a.shape, a.strides):
        if st < 0:
            lo += (n - 1) * st
        else:
            hi += (n - 1) * st
    return lo, hi + a.itemsize


class DirtySlots:
    """Dirty slot ranges of a ring: a short sorted list of disjoint [lo, hi) intervals (touching / overlapping ones merge;
    beyond MAX intervals the two closest neighbours are joined — an upload that is a bit too large costs little DMA time, a
    missed slot would be a wrong result)."""
    MAX = 32

    def __init__(self):
        self.iv = []

    def mark(self, lo, hi):
        if hi <= lo:
            return
        out, placed = [], False
        for a, b in self.iv:
            if b < lo or hi < a:                          # disjoint and not touching
                if not placed and hi < a:
                    out.append((lo, hi)); placed = True
                out.append((a, b))
            else:
                lo, hi = min(lo, a), max(hi, b)
        if not placed:
            out.append((lo, hi))
        out.sort()
        while len(out) > self.MAX:
            k = min(range(len(out) - 1), key=lambda i: out[i + 1][0] - out[i][1])
            out[k:k + 2] = [(out[k][0], out[k + 1][1])]
        self.iv = out

    def take(self):
        r, self.iv = self.iv, []
        return r

    @property
    def lo(self):
        return self.iv[0][0] if self.iv else None

    @property
    def hi(self):
        return self.iv[-1][1] if self.iv else None

    def __bool__(self):
        return bool(self.iv)
