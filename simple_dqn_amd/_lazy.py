"""Lazy host view of the gathered states (round 5; reference call pattern `net.train(mem.getMinibatch(), epoch)`, agent.py:112-114).

getMinibatch() of the reference returns five numpy arrays (replay_memory.py:79).  Here the gather runs on the GPU, and most callers hand
the tuple straight back to DeepQNetwork.train without ever reading the 2 x B x 28 KB of states on the host — copying them down and
waiting for the copy on every call made the reference's own loop body run at 44 % of the fused loop's rate.  `LazyMinibatchArray` stands
for `mem.prestates` / `mem.poststates` in the returned tuple: shape / dtype / len are known at once, ANY look at the data (indexing,
iteration, numpy functions and ufuncs via `__array__`, every ndarray method or attribute, comparison and arithmetic operators, repr)
first fetches the device minibatch into the pinned host buffers (`ReplayMemory._materialize`: one D2H + wait, once per gather) and
then behaves like the ndarray `mem.prestates` itself — same memory, same aliasing as the reference (the buffers are re-used by the
next getMinibatch(), replay_memory.py:21-22,76-77), same write tracking.

CAVEAT (differs from the reference): only the lazy OBJECT is live.  An ndarray taken from it earlier — `p = np.asarray(
mem.prestates)` — is the pinned buffer itself, but the next getMinibatch() no longer fills that buffer at once: `p` keeps showing the
previous batch until somebody looks at the lazy object again (any access fetches), whereas the reference's `mem.prestates` would already
show the new one.  Code that holds such an alias across getMinibatch() calls (statistics.py:85-86 keeps `mem.prestates` itself, which IS
the live object) must touch the lazy object — or call `mem._materialize()` — before reading the alias.

There is ONE such object per ReplayMemory and role; it is what the `prestates` / `poststates` attributes are, so `pre is mem.prestates`
holds for the tuple's elements exactly as in the reference (:79 returns the attributes themselves).

It is deliberately NOT an ndarray subclass: consumers that read an ndarray's memory without calling into Python (the buffer protocol,
C extensions) cannot reach the pinned buffer through this object before it has been filled — `memoryview(x)` raises TypeError instead of
showing stale bytes; `np.asarray(x)` gives the real array.
"""
import weakref

import numpy as np


class LazyMinibatchArray:
    __slots__ = ("_memref", "_which", "_shape", "__weakref__")
    __array_priority__ = 0.0

    def __init__(self, mem, which):
        self._memref, self._which = weakref.ref(mem), which       # (one per ReplayMemory and role, owned by it: no reference cycle)
        self._shape = (mem.batch_size, mem.history_length) + tuple(mem.dims)

    @property
    def _mem(self):
        m = self._memref()
        if m is None:
            raise ReferenceError("the ReplayMemory this minibatch array belongs to has been destroyed")
        return m

    # ---- known without the data ----------------------------------------------------------------------------------------
    @property
    def shape(self):
        return self._shape

    dtype = np.dtype(np.uint8)
    ndim = 4

    @property
    def size(self):
        return int(np.prod(self._shape))

    @property
    def nbytes(self):
        return self.size

    def __len__(self):
        return self._shape[0]

    # ---- everything else looks at the data ---------------------------------------------------------------------------------
    def _arr(self):
        return self._mem._states(self._which)              # materialises the host copy if the last gather has not been fetched yet

    def __array__(self, dtype=None, copy=None):
        a = self._arr()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __getitem__(self, key):
        return self._arr()[key]

    def __setitem__(self, key, value):
        self._arr()[key] = value

    def __iter__(self):
        return iter(self._arr())

    def __getattr__(self, name):                           # (only reached for names not defined here: every ndarray method / attribute)
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self._arr(), name)

    def __repr__(self):
        return repr(self._arr())

    def __str__(self):
        return str(self._arr())

    def __bool__(self):
        return bool(self._arr())

    def __contains__(self, x):
        return x in self._arr()

    def __copy__(self):
        return self._arr().copy()

    def __deepcopy__(self, memo):
        return self._arr().copy()

    def __reduce__(self):                                  # pickling stores the data, not the handle
        return (np.array, (np.asarray(self._arr()).copy(),))


def _binary(name):
    def op(self, other):
        if isinstance(other, LazyMinibatchArray):
            other = other._arr()
        return getattr(self._arr(), name)(other)
    op.__name__ = name
    return op


def _unary(name):
    def op(self):
        return getattr(self._arr(), name)()
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "floordiv", "mod", "pow", "and", "or", "xor", "lshift", "rshift", "matmul"):
    for _pre in ("__%s__", "__r%s__", "__i%s__"):
        setattr(LazyMinibatchArray, _pre % _n, _binary(_pre % _n))
for _n in ("__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__divmod__", "__rdivmod__"):
    setattr(LazyMinibatchArray, _n, _binary(_n))
for _n in ("__neg__", "__pos__", "__abs__", "__invert__"):
    setattr(LazyMinibatchArray, _n, _unary(_n))
LazyMinibatchArray.__hash__ = None                         # (like ndarray: == is elementwise)
