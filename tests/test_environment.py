"""Environment interface (src/environment.py:7-33) — synthetic implementation and the optional gym adapter."""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_dqn_amd.environment import GymEnvironment, SyntheticEnvironment, _to_gray_resized  # noqa: E402
from util import make_args  # noqa: E402


class _FakeEnv:
    def __init__(self, new_api):
        self.new_api = new_api
        self.action_space = types.SimpleNamespace(n=6)
        self.t = 0

    def _obs(self):
        return np.full((210, 160, 3), (self.t * 10) % 256, dtype=np.uint8)

    def reset(self):
        self.t = 0
        return (self._obs(), {}) if self.new_api else self._obs()

    def step(self, a):
        self.t += 1
        done = self.t >= 3
        return (self._obs(), 1.0, done, False, {}) if self.new_api else (self._obs(), 1.0, done, {})


def test_gym_adapter_both_api_generations():
    args = make_args()
    for new_api in (False, True):
        env = GymEnvironment("Breakout-v0", args, make=lambda _id, n=new_api: _FakeEnv(n))
        assert env.numActions() == 6
        env.restart()
        assert env.isTerminal() is False and env.getScreen().shape == (84, 84) and env.getScreen().dtype == np.uint8
        rewards = [env.act(0) for _ in range(3)]
        assert rewards == [1.0, 1.0, 1.0] and env.isTerminal() is True
        assert int(env.getScreen()[10, 10]) == 30                       # constant frames survive luma + resize exactly


def test_gray_resize_properties():
    rng = np.random.RandomState(0)
    g = rng.randint(0, 256, (84, 84)).astype(np.uint8)
    assert np.array_equal(_to_gray_resized(g, 84, 84), g)               # identity at the native size
    ramp = np.tile(np.arange(160, dtype=np.float64), (210, 1))
    r = _to_gray_resized(ramp, 84, 84)
    assert np.all(np.diff(r.astype(int), axis=1) >= 0) and r[:, 0].max() <= 1 and r[:, -1].min() >= 158   # monotone, end points kept
    rgb = np.zeros((4, 4, 3), np.uint8); rgb[..., 1] = 255
    assert int(_to_gray_resized(rgb, 4, 4)[0, 0]) == 150                # 0.587 * 255


def test_synthetic_environment_interface():
    env = SyntheticEnvironment(make_args(), num_actions=4, seed=1)
    env.restart()
    assert env.numActions() == 4 and env.getScreen().shape == (84, 84) and env.getScreen().dtype == np.uint8
    r = env.act(2)
    assert r in (-1, 0, 1) and isinstance(env.isTerminal(), bool)
