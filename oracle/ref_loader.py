"""Loader of the reference's own ReplayMemory class — test/bench infrastructure, NOT product code.

Prefers the live source under /root/reference (this container), else the byte-compiled copy oracle/_ref/replay_memory.pyc
made by oracle/build_ref.py (the GPU box).  The module is loaded under a private name with an explicit path: sys.path is
never touched, so the reference's `statistics` / `util` cannot shadow anything.  Returns None when neither exists."""
import importlib.machinery
import importlib.util
import os
import sys
import warnings

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src/replay_memory.py"
REF_PYC = os.path.join(_HERE, "_ref", "replay_memory.pyc")
_NAME = "_sdqn_reference_replay_memory"


def load_reference_replay_memory():
    """-> (ReplayMemory class of the reference, provenance string) or (None, reason)."""
    if _NAME in sys.modules:
        m = sys.modules[_NAME]
        return m.ReplayMemory, m.__sdqn_provenance__
    if os.path.exists(REF_SRC):
        loader, origin = importlib.machinery.SourceFileLoader(_NAME, REF_SRC), "live source %s" % REF_SRC
    elif os.path.exists(REF_PYC):
        loader, origin = importlib.machinery.SourcelessFileLoader(_NAME, REF_PYC), \
            "byte-compiled from /root/reference/src/replay_memory.py by oracle/build_ref.py (oracle/_ref/replay_memory.pyc)"
    else:
        return None, "neither /root/reference nor oracle/_ref/replay_memory.pyc present"
    spec = importlib.util.spec_from_loader(_NAME, loader)
    mod = importlib.util.module_from_spec(spec)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # np.integer dtype DeprecationWarning of the 2015 code
            loader.exec_module(mod)
    except Exception as e:                           # e.g. a .pyc of another CPython version (bad magic number)
        return None, "reference module failed to load: %r" % (e,)
    mod.__sdqn_provenance__ = origin
    sys.modules[_NAME] = mod
    return mod.ReplayMemory, origin
