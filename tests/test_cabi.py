"""CPU-side checks of the C ABI: the library loads, exports every symbol include/sdqn.h declares,
the native sampler is bit-exact against CPython's random stream and the reference KATs, and device
entry points fail loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import json
import os
import random
import re

import numpy as np
import pytest

import simple_dqn_amd as sd
from simple_dqn_amd import _lib
from oracle.replay_numpy import ReplayOracle, synthetic_fill

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "sdqn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sdqn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    lib = sd.load()
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.sdqn_version() >= 100


def test_cfg_struct_layout():
    assert C.sizeof(_lib.NetCfg) == 8 * 4 + 11 * 8
    assert _lib.NetCfg.optimizer.offset == 24 and _lib.NetCfg.datatype.offset == 28 and _lib.NetCfg.beta_1.offset == 32 + 7 * 8


def test_mt_seed_and_randint_bit_exact():
    lib = sd.load()
    mt = (C.c_uint32 * 625)()
    out = C.c_int64()
    for seed in (0, 42, 1234, 2**31 + 5, 2**40 + 17):
        random.seed(seed)
        _lib.check(lib.sdqn_mt_seed(mt, seed))
        assert tuple(mt) == random.getstate()[1]
        for n in (2, 7, 9996, 2**20 + 1, 2**32 - 5, 2**33 + 11):
            for _ in range(40):
                _lib.check(lib.sdqn_mt_randint(mt, 4, 3 + n, C.byref(out)))
                assert out.value == random.randint(4, 3 + n)
        assert tuple(mt) == random.getstate()[1]


def test_native_sampler_matches_reference_kats(golden_dir):
    lib = sd.load()
    kats = json.load(open(os.path.join(golden_dir, "replay_kat.json")))["kats"]
    for k in kats:
        m = ReplayOracle(k["size"], batch_size=k["B"])
        synthetic_fill(m, k["fill_seed"], count=k["count"], current=k["current"])
        term = np.ascontiguousarray(m.terminals.view(np.uint8))
        mt = (C.c_uint32 * 625)()
        _lib.check(lib.sdqn_mt_seed(mt, k["seed"]))
        idx = np.empty(k["B"], np.int64)
        draws = C.c_int64()
        for call in k["calls"]:
            _lib.check(lib.sdqn_sample_indices(mt, _lib.ptr(term, C.c_uint8), k["count"], k["current"], 4, k["B"],
                                               _lib.ptr(idx, C.c_int64), C.byref(draws)))
            assert idx.tolist() == call["indexes"] and draws.value == call["draws"]


def test_generator_resync_by_words_drawn():
    """train_from_memory hands the library a COPY of random.getstate()[1] and afterwards advances Python's own generator by the number
    of 32-bit words the library drew (sdqn_mt_words): random.getrandbits(32 * n) consumes exactly n words (ceil(k / 32) genrand_uint32
    calls), so Python's state equals the library's copy — for ring sizes below and above 2^32 / with rejections, across a twist."""
    import array
    lib = sd.load()
    term = np.zeros(50000, np.uint8); term[::97] = 1
    idx = np.empty(32, np.int64)
    random.seed(77)
    for rep in range(60):
        random.random(); random.randrange(6)                         # the agent's own draws in between (agent.py:50-51)
        st = random.getstate()
        arr = array.array("I", st[1])
        mt = (C.c_uint32 * 625).from_buffer(arr)
        w0, w1 = C.c_uint64(), C.c_uint64()
        _lib.check(lib.sdqn_mt_words(C.byref(w0)))
        _lib.check(lib.sdqn_sample_indices(mt, _lib.ptr(term, C.c_uint8), 50000 - rep, 1234, 4, 32, _lib.ptr(idx, C.c_int64), None))
        _lib.check(lib.sdqn_mt_words(C.byref(w1)))
        n = w1.value - w0.value
        assert n >= 32
        random.getrandbits(32 * n)
        assert random.getstate()[1] == tuple(arr)


def test_sampler_preconditions():
    lib = sd.load()
    mt = (C.c_uint32 * 625)()
    lib.sdqn_mt_seed(mt, 1)
    term = np.zeros(100, np.uint8)
    idx = np.empty(8, np.int64)
    with pytest.raises(AssertionError):        # replay_memory.py:52 assert count > history_length
        _lib.check(lib.sdqn_sample_indices(mt, _lib.ptr(term, C.c_uint8), 4, 0, 4, 8, _lib.ptr(idx, C.c_int64), None))
    term[:] = 1
    with pytest.raises(AssertionError):        # no admissible index: the reference would spin forever
        _lib.check(lib.sdqn_sample_indices(mt, _lib.ptr(term, C.c_uint8), 100, 0, 4, 8, _lib.ptr(idx, C.c_int64), None))


def test_no_cpu_fallback():
    lib = sd.load()
    n = C.c_int(0)
    rc = lib.sdqn_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    from util import make_args
    with pytest.raises(sd.SdqnError):
        sd.ReplayMemory(100, make_args())
    with pytest.raises(sd.SdqnError):
        sd.DeepQNetwork(4, make_args())


def test_product_never_imports_oracle():
    pk = os.path.join(ROOT, "simple_dqn_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("oracle/", "ORACLE_DIR").lower() or "import oracle" not in src, f
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_layer_shapes_follow_the_input_geometry():
    """deepqnetwork.py:83-91 builds the same stack for any --screen_width / --screen_height / --history_length (main.py:27-28,34); the
    drop-in's Neon-layout shapes must be the oracle's for every geometry, and a screen the stack cannot digest is an AssertionError."""
    import pytest
    from simple_dqn_amd.deepqnetwork import layer_shapes
    from oracle.dqn_numpy import layer_shapes as oracle_shapes
    assert layer_shapes(4) == [(256, 32), (512, 64), (576, 64), (512, 3136), (4, 512)]
    for A, hist, H, W in ((6, 4, 84, 84), (3, 2, 64, 48), (18, 5, 36, 36), (4, 1, 210, 160), (2, 3, 60, 52)):
        assert layer_shapes(A, hist, H, W) == oracle_shapes(A, hist, H, W)
    with pytest.raises(AssertionError):
        layer_shapes(4, 4, 20, 20)
