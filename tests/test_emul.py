"""Runs the problem definitions of simple_dqn_amd/csrc/problems.h (the index math the HIP tile engine
executes) on the host through tests/emul/emul.cpp and compares with the oracle: validates every
im2col/dgrad/wgrad gather, the padded-delta layouts and the Neon<->internal layout converters on CPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.dqn_numpy import OracleDQN, xavier_weights
from util import random_minibatch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(HERE, "emul", "libsdqn_emul.so")
    src = os.path.join(HERE, "emul", "emul.cpp")
    hdr = os.path.join(HERE, "..", "simple_dqn_amd", "csrc", "problems.h")
    hdr2 = os.path.join(HERE, "..", "simple_dqn_amd", "csrc", "bt_map.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(hdr2)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src])
    return C.CDLL(so)


@pytest.mark.parametrize("B,A,seed,bt", [(4, 6, 5, 0), (3, 4, 8, 0), (5, 18, 2, 0), (3, 4, 9, 1), (5, 6, 3, 2)])
def test_problem_index_math(emul, B, A, seed, bt):
    """bt = 0: the problem structs with naive loops.  bt = 1 / 2: the same step through the BLOCK-TILE engine's maps (bt_map.h: loader
    items, the two LDS panel layouts, fragment offsets, accumulator rows) on a simulated workgroup, at the built-in block shapes and at
    the alternative ones of the menu — small B leaves ragged blocks in every stage (M = 81 B, 49 B, B ...)."""
    emul.emul_set_bt(bt)
    ws, wt = xavier_weights(A, seed), xavier_weights(A, seed + 1)
    o = OracleDQN(A, batch_size=B, weights=ws)
    o.Wt = [w.copy() for w in wt]
    pre, act, rew, post, term = random_minibatch(B, A, seed + 2, p_term=0.3, reward_range=(-3, 4))
    g, cost, deltas, preq = o.gradients((pre, act, rew, post, term))
    fp = C.POINTER(C.c_float)
    arr = lambda lst: (fp * 5)(*[w.ctypes.data_as(fp) for w in lst])
    gout = [np.zeros_like(w) for w in ws]
    q = np.zeros((2, B, A), np.float32)
    cst = C.c_float()
    t8 = term.astype(np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    emul.emul_step(B, A, arr(ws), arr(wt), vp(pre), vp(act), vp(rew), vp(post), vp(t8), C.c_double(0.99),
                   C.c_double(1.0), C.c_double(-1.0), C.c_double(1.0), q.ctypes.data_as(fp), arr(gout), C.byref(cst))
    emul.emul_set_bt(0)
    assert np.abs(q[0] - preq).max() < 1e-5
    assert abs(cst.value - float(cost)) < 1e-5
    for i in range(5):
        assert np.abs(gout[i] - g[i]).max() < 2e-5 * max(1.0, np.abs(g[i]).max()), i


def test_normalisation_is_exact_division(emul):
    # deepqnetwork.py:100 be.divide(input, 255): the division-free device formula must equal IEEE x/255
    out = np.zeros(256, np.float32)
    emul.emul_norm_u8(out.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(out, np.arange(256, dtype=np.float32) / np.float32(255))


def test_batch_size_division_is_exact(emul):
    """A9's grad / be.bsz: the power-of-two shortcut (multiply by the exact reciprocal) must equal IEEE division for
    every input, and other divisors must take the division path."""
    rng = np.random.RandomState(5)
    x = rng.randint(0, 2 ** 32, size=2_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x = x[np.isfinite(x)]
    out = np.empty_like(x)
    emul.emul_div_bsz.argtypes = [C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.c_int]
    for bsz in (1.0, 2.0, 32.0, 64.0, 256.0, 2048.0, 3.0, 96.0, 0.5):
        emul.emul_div_bsz(x.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(bsz), out.ctypes.data_as(C.POINTER(C.c_float)), len(x))
        with np.errstate(all="ignore"):
            ref = x / np.float32(bsz)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), bsz


def test_xcd_contiguous_maps_are_bijections_with_contiguous_runs(emul):
    """problems.h: xcd_tile_id / xcd_tile_id_range (the placement maps of every launch that shares operands inside an XCD's L2 — round 4:
    the B >= 128 backward launches, the fc4_wgrad tiles of bwd3).  Workgroup b runs on XCD b mod 8: for every grid size, and for every
    sub-range [s, s + n) of a multi-problem launch, the map must hit every tile exactly once and give each XCD one contiguous run."""
    for nwg in list(range(1, 70)) + [98, 162, 196, 200, 324, 392, 502, 648, 800, 1568, 4016]:
        t = [emul.emul_xcd_tile_id(b, nwg) for b in range(nwg)]
        assert sorted(t) == list(range(nwg)), nwg
        for x in range(8):
            run = sorted(t[b] for b in range(x, nwg, 8))
            assert run == list(range(run[0], run[0] + len(run))) if run else True, (nwg, x)
    for s0 in (0, 1, 3, 7, 8, 13, 162, 306, 324):
        for n in list(range(1, 40)) + [98, 144, 196, 352, 400, 800]:
            t = [emul.emul_xcd_tile_id_range(b, s0, n) for b in range(s0, s0 + n)]
            assert sorted(t) == list(range(n)), (s0, n)
            for x in range(8):
                run = sorted(t[b - s0] for b in range(s0, s0 + n) if b % 8 == x)
                assert (not run) or run == list(range(run[0], run[0] + len(run))), (s0, n, x)

