"""Neon snapshot reader/writer against synthetic pickles in both recalled schemas."""
import os
import pickle
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_dqn_amd.neon_compat import LAYER_SHAPES, read_neon_pickle, write_neon_pickle  # noqa: E402


def _weights(A, seed):
    rng = np.random.RandomState(seed)
    return [rng.uniform(-1, 1, size=s).astype(np.float32) for s in LAYER_SHAPES(A)]


@pytest.mark.parametrize("wrapped", [False, True])
def test_old_schema_layer_params_states(tmp_path, wrapped):
    # neon 1.0/1.1 Model.serialize; `wrapped` = after /root/reference/src/util/convert_weights.py:11-13
    ws, ss = _weights(4, 1), _weights(4, 2)
    d = {"epoch_index": 7, "layer_params_states": [{"params": ({"W": w} if wrapped else w), "states": [s]} for w, s in zip(ws, ss)]}
    p = tmp_path / "old.pkl"
    with open(p, "wb") as f:
        pickle.dump(d, f, protocol=2)
    rw, rs, A = read_neon_pickle(str(p))
    assert A == 4
    for a, b in zip(rw, ws):
        assert np.array_equal(a, b)
    for a, b in zip(rs, ss):
        assert np.array_equal(a, b)


def test_new_schema_nested_containers_and_missing_states(tmp_path):
    ws = _weights(6, 3)
    wl = [{"type": "neon.layers.layer.Convolution", "config": {}, "params": {"W": w}} for w in ws[:3]] + \
         [{"type": "neon.layers.layer.Linear", "config": {}, "params": {"W": w}, "states": []} for w in ws[3:]]
    act = {"type": "neon.layers.layer.Activation", "config": {}}
    layers = [wl[0], act, wl[1], act, {"type": "neon.layers.container.Sequential", "config": {"layers": [wl[2], act, wl[3], act]}}, wl[4]]
    d = {"model": {"type": "neon.models.model.Model", "config": {"layers": layers}}, "epoch_index": 0}
    p = tmp_path / "new.pkl"
    with open(p, "wb") as f:
        pickle.dump(d, f)
    rw, rs, A = read_neon_pickle(str(p))
    assert A == 6 and rs is None
    for a, b in zip(rw, ws):
        assert np.array_equal(a, b)


def test_writer_roundtrip_and_rejects_other_architectures(tmp_path):
    ws, ss = _weights(18, 4), _weights(18, 5)
    p = tmp_path / "out.prm"
    write_neon_pickle(str(p), ws, ss, epoch_index=3)
    rw, rs, A = read_neon_pickle(str(p))
    assert A == 18
    for a, b in zip(rw + rs, ws + ss):
        assert np.array_equal(a, b)
    bad = {"layer_params_states": [{"params": np.zeros((3, 3), np.float32)}] * 5}
    q = tmp_path / "bad.pkl"
    with open(q, "wb") as f:
        pickle.dump(bad, f)
    with pytest.raises(ValueError):
        read_neon_pickle(str(q))
    with open(q, "wb") as f:
        pickle.dump({"layer_params_states": []}, f)
    with pytest.raises(ValueError):
        read_neon_pickle(str(q))
