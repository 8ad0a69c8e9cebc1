"""simple_dqn_amd — MI355X (gfx950) native DQN training step behind simple_dqn's Python API.

Drop-in replacements for the reference's hot-path classes (tambetm/simple_dqn):
    ReplayMemory   <- src/replay_memory.py:6-79
    DeepQNetwork   <- src/deepqnetwork.py:15-192
    StateBuffer    <- src/state_buffer.py:3-27
    Agent          <- src/agent.py:7-135
All arithmetic runs in hand-written HIP kernels inside libsdqn_hip.so (C ABI: include/sdqn.h),
reached through ctypes (_lib.py).  There is no CPU fallback: importing works anywhere, but
constructing a ReplayMemory / DeepQNetwork without a HIP device raises RuntimeError.
"""
from ._lib import lib_path, load, SdqnError  # noqa: F401
from .replay_memory import ReplayMemory  # noqa: F401
from .deepqnetwork import DeepQNetwork  # noqa: F401
from .state_buffer import DeviceStateBuffer, StateBuffer  # noqa: F401
from .agent import Agent  # noqa: F401
from .environment import SyntheticEnvironment  # noqa: F401
from .statistics import Statistics  # noqa: F401

__all__ = ["ReplayMemory", "DeepQNetwork", "StateBuffer", "DeviceStateBuffer", "Agent", "SyntheticEnvironment", "Statistics", "load", "lib_path"]
