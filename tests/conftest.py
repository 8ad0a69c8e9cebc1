import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _device_count():
    """Devices the LIBRARY sees.  Only 'no ROCm device here' may turn into skips: without /dev/kfd (this container) the
    answer is 0; with a GPU node present every load / symbol / runtime error propagates, so a broken build on the GPU box
    shows up as errors, never as a green run full of skips."""
    if not os.path.exists("/dev/kfd"):
        return 0
    import ctypes as C
    import simple_dqn_amd as sd
    lib = sd.load()                                   # missing .so / header-library mismatch: raises
    n = C.c_int(0)
    rc = lib.sdqn_device_count(C.byref(n))
    if rc != 0:
        raise RuntimeError("sdqn_device_count failed on a box with /dev/kfd: %s" % (lib.sdqn_last_error() or b"").decode())
    return n.value


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a ROCm device: on a box without one they are skipped (with the reason), not failed.  On a GPU
    box nothing is skipped and an unloadable libsdqn_hip.so aborts the collection loudly (_device_count)."""
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items or _device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no ROCm device visible (libsdqn_hip has no CPU path)")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """Build libsdqn_hip.so in-tree if it is missing or older than its sources (hipcc cross-compiles gfx950 without a
    GPU).  Tests never fall back to anything else: a failed build surfaces as failing tests."""
    import subprocess
    csrc = os.path.join(ROOT, "simple_dqn_amd", "csrc")
    so = os.path.join(ROOT, "simple_dqn_amd", "libsdqn_hip.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip"))] + \
           [os.path.join(ROOT, "include", "sdqn.h")]
    stale = not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs)
    if stale and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.call(["make", "-C", csrc, "-j4"])
