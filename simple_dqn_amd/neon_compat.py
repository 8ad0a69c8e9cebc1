"""Best-effort reader/writer for Neon weight pickles.

The reference saves and loads snapshots with `model.save_params(path)` / `model.load_params(path)`
(/root/reference/src/deepqnetwork.py:188-192), i.e. whatever pickle its Neon version writes.  Neon is not
vendored or pinned by the reference and none of its snapshots ship with it, so the two schemas
below are [neon-recalled] and exercised only against synthetic pickles (tests/test_neon_compat.py):

* old (neon 1.0/1.1 `Model.serialize`):  {'epoch_index': n, 'layer_params_states': [{'params': W | {'W': W},
  'states': [s, ...]}, ...]} with one entry per weight layer — the variant /root/reference/src/util/convert_weights.py:11-13
  converts between (bare array -> {'W': array});
* new (neon >= 1.2 `get_description(get_weights=True)` / `save_params`): {'model': {'type': ..., 'config': {'layers':
  [{'type': 'neon.layers.layer.Convolution', 'config': {...}, 'params': {'W': W}, 'states': [...]}, ...]}}, ...},
  containers nest further `config.layers` lists.

Layouts need no conversion: Neon conv weights are (C*R*S, K), Linear weights (nout, nin) — exactly the
"Neon layout" of the C ABI (include/sdqn.h, sdqn_net_set_weights).
"""
import pickle

import numpy as np

LAYER_SHAPES = lambda num_actions: [(256, 32), (512, 64), (576, 64), (512, 3136), (num_actions, 512)]   # noqa: E731


def _as_array(w):
    if isinstance(w, dict):
        w = w.get("W")
    if w is None:
        return None
    a = np.asarray(w)
    return a if a.ndim == 2 else None


def _walk(node, out):
    """Depth-first, in order: collect (W, states) of every dict that carries weights under 'params'."""
    if isinstance(node, dict):
        if "params" in node:
            w = _as_array(node["params"])
            if w is not None:
                st = node.get("states") or []
                out.append((w, [np.asarray(s) for s in st if s is not None]))
                return
        for key in ("model", "config", "layers", "layer_params_states"):
            if key in node:
                _walk(node[key], out)
    elif isinstance(node, (list, tuple)):
        for x in node:
            _walk(x, out)


def read_neon_pickle(path):
    """-> (weights: list of 5 float32 arrays in Neon layout, states: list of 5 arrays or None, num_actions).
    Raises ValueError when the file does not describe the DQN architecture of deepqnetwork.py:77-92."""
    with open(path, "rb") as f:
        try:
            d = pickle.load(f)
        except UnicodeDecodeError:                     # python-2 pickle holding numpy arrays
            f.seek(0)
            d = pickle.load(f, encoding="latin1")
    found = []
    _walk(d, found)
    if len(found) != 5:
        raise ValueError("expected 5 weight layers (3 conv + 2 affine) in %s, found %d" % (path, len(found)))
    num_actions = int(found[4][0].shape[0])
    ws, sts = [], []
    for i, ((w, st), shape) in enumerate(zip(found, LAYER_SHAPES(num_actions))):
        if tuple(w.shape) != shape:
            raise ValueError("layer %d has shape %s, the DQN network needs %s" % (i, tuple(w.shape), shape))
        ws.append(np.ascontiguousarray(w, dtype=np.float32))
        s = next((x for x in st if tuple(x.shape) == shape), None)
        sts.append(None if s is None else np.ascontiguousarray(s, dtype=np.float32))
    states = sts if all(s is not None for s in sts) else None
    return ws, states, num_actions


def write_neon_pickle(path, weights, states=None, epoch_index=0):
    """Writes the newer description schema (what `Model.load_params` of neon >= 1.2 reads [neon-recalled]):
    Conv/Affine compound layers of deepqnetwork.py:83-91 expanded to Convolution|Linear + Activation entries."""
    def conv(fshape, stride):
        return {"type": "neon.layers.layer.Convolution", "config": {"fshape": fshape, "strides": stride, "padding": 0}}

    def lin(nout):
        return {"type": "neon.layers.layer.Linear", "config": {"nout": nout}}

    def act(name):
        return {"type": "neon.layers.layer.Activation", "config": {"transform": {"type": "neon.transforms.activation." + name, "config": {}}}}

    A = int(np.asarray(weights[4]).shape[0])
    protos = [conv((8, 8, 32), 4), conv((4, 4, 64), 2), conv((3, 3, 64), 1), lin(512), lin(A)]
    layers = []
    for i, proto in enumerate(protos):
        proto["params"] = {"W": np.asarray(weights[i], dtype=np.float32)}
        proto["states"] = [np.asarray(states[i], dtype=np.float32)] if states is not None else []
        layers.append(proto)
        if i < 4:
            layers.append(act("Rectlin"))
    d = {"epoch_index": int(epoch_index), "neon_version": "simple_dqn_amd-export",
         "model": {"type": "neon.models.model.Model",
                   "config": {"layers": layers}}}
    with open(path, "wb") as f:
        pickle.dump(d, f, protocol=2)
