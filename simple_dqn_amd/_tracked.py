"""Write-tracking numpy views of the replay ring.

The reference keeps ONE copy of the ring (replay_memory.py:10-13): whatever a caller writes into `mem.screens`,
`mem.actions`, `mem.rewards` or `mem.terminals` is what the next getMinibatch() reads.  Here those attributes are views of
the pinned host master copy and the kernels read an HBM mirror, so a direct write must reach the mirror before the next
gather.  `TrackedArray` is an ndarray subclass whose every in-place write path records WHICH ring slots it touched in the
owning ReplayMemory (a dirty slot range); the memory uploads exactly that range before its next device use.  Nothing is
ever trained on stale frames silently, and no `sync_mirror()` call is needed for correctness any more (it remains as an
explicit bulk upload).

Every array the class hands out is flagged READ-ONLY for numpy; the tracked write paths below go through a private writable alias
of the same memory (`_w()`).  So an alias that escapes the tracking — `np.asarray(mem.screens)`, `.view(np.ndarray)`,
`memoryview(...)`, `torch.from_numpy(...)` — cannot be written at all: `x = np.asarray(mem.screens); x[5] = 7` raises
"assignment destination is read-only" instead of training on a stale HBM mirror silently.  A caller who
re-enables writing on such an alias (`x.flags.writeable = True`) or writes through raw pointers owns the `sync_mirror()` call.

Write paths covered: indexing assignment (`a[...] = v`, any index kind), in-place operators and every ufunc / numpy
function with `out=` (`np.bitwise_xor(x, y, out=a[s:e])`), `np.copyto / put / place / putmask / put_along_axis`, and the
mutating ndarray methods (`fill`, `put`, `sort`, `partition`, `byteswap(inplace=True)`, `setfield`, `itemset`).  Views of
views stay tracked (`a[10:20][3] = v`), also after `.view(dtype)` / `.reshape` of the same memory.  NOT covered: writes
through ctypes / raw pointers — those callers own their `sync_mirror()`.
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
        obj.flags.writeable = False             # every view / alias derived from this object inherits the flag
        return obj

    def _w(self):
        """Writable plain-ndarray alias of exactly this view (same memory, shape, strides) — the only door for writes."""
        if self.flags.writeable:                # a copy that owns its data (fancy-index result, .copy()): nothing to track
            return self.view(np.ndarray)
        ref = self._owner
        owner = ref() if ref is not None else None
        raw = owner._raw_bytes.get(self._kind) if owner is not None else None
        if raw is None or self.size == 0:
            return self.view(np.ndarray)        # (read-only: a write raises)
        base = raw.__array_interface__["data"][0]
        lo, hi = _byte_bounds(self)
        if lo < base or hi > base + raw.nbytes:
            return self.view(np.ndarray)
        return np.ndarray(self.shape, dtype=self.dtype, buffer=raw, offset=self.__array_interface__["data"][0] - base, strides=self.strides)

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
        k0 = key[0] if isinstance(key, tuple) and key else key
        try:
            if isinstance(k0, (int, np.integer, slice)):
                target = np.ndarray.__getitem__(self, k0)
        except Exception:
            target = self
        if isinstance(value, TrackedArray):
            value = value.view(np.ndarray)
        self._w()[key] = value
        if self.flags.writeable:                # (an untracked copy)
            return
        if isinstance(k0, (list, np.ndarray)) and self.ndim >= 1 and self._touched_rows(k0):
            return                              # integer-array / boolean-mask key on the first axis: only the rows it names
        if isinstance(target, TrackedArray):
            target._touched()
        else:                                   # a 0-d element (a[i] = v on a 1-d array): one slot
            self._touched_element(key)

    def _touched_rows(self, k0):
        """Fancy key on axis 0: mark the named rows (each one while they are few, their span otherwise) instead of the
        whole view — `mem.screens[[i, j, k]] = frames` on a 1 M-frame ring must not schedule a 7 GB upload."""
        try:
            idx = np.asarray(k0)
            if idx.dtype == np.bool_:
                if idx.ndim != 1 or idx.shape[0] != self.shape[0]:
                    return False
                idx = np.flatnonzero(idx)
            elif not np.issubdtype(idx.dtype, np.integer):
                return False
            idx = idx.reshape(-1).astype(np.int64)
            if idx.size == 0:
                return True
            idx = np.where(idx < 0, idx + self.shape[0], idx)
            rows = np.unique(idx) if idx.size <= 64 else np.array([idx.min(), idx.max()])
            if idx.size <= 64:
                for i in rows:
                    np.ndarray.__getitem__(self, slice(int(i), int(i) + 1))._touched()
            else:
                np.ndarray.__getitem__(self, slice(int(rows[0]), int(rows[1]) + 1))._touched()
            return True
        except Exception:
            return False

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
        if method == "at" and inputs and isinstance(inputs[0], TrackedArray):
            inputs = (inputs[0]._w(),) + tuple(inputs[1:])
        inputs = tuple(x.view(np.ndarray) if isinstance(x, TrackedArray) else x for x in inputs)
        if outs:
            kwargs["out"] = tuple(o._w() if isinstance(o, TrackedArray) else o for o in outs)
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
        args, kwargs = list(args), dict(kwargs)
        if name in _MUTATING_FUNCS:
            pname = _MUTATING_FUNCS[name]
            if args and isinstance(args[0], TrackedArray):
                written.append(args[0]); args[0] = args[0]._w()
            elif isinstance(kwargs.get(pname), TrackedArray):
                written.append(kwargs[pname]); kwargs[pname] = kwargs[pname]._w()
        o = kwargs.get("out")
        if isinstance(o, tuple):
            written += [x for x in o if isinstance(x, TrackedArray)]
            kwargs["out"] = tuple(x._w() if isinstance(x, TrackedArray) else x for x in o)
        elif isinstance(o, TrackedArray):
            written.append(o); kwargs["out"] = o._w()
        if not written:
            return super().__array_function__(func, types, tuple(args), kwargs)
        # the write targets are plain writable aliases now; the remaining tracked arguments are read-only inputs
        plain = lambda x: x.view(np.ndarray) if isinstance(x, TrackedArray) else x
        res = func(*[plain(x) for x in args], **{k: plain(v) for k, v in kwargs.items()})
        for w in written:
            w._touched()
        return res

    def fill(self, value):
        self._w().fill(value); self._touched()

    def put(self, *a, **k):
        self._w().put(*a, **k); self._touched()

    def sort(self, *a, **k):
        self._w().sort(*a, **k); self._touched()

    def partition(self, *a, **k):
        self._w().partition(*a, **k); self._touched()

    def setfield(self, *a, **k):
        self._w().setfield(*a, **k); self._touched()

    def byteswap(self, inplace=False):
        if not inplace:
            return np.ndarray.byteswap(self.view(np.ndarray), False)
        self._w().byteswap(True); self._touched()
        return self

    def itemset(self, *a):                      # (numpy < 2)
        self._w().itemset(*a); self._touched()


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
