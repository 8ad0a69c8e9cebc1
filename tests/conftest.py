import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
