"""Environment interface of /root/reference/src/environment.py:7-33 plus a synthetic implementation.

The reference's ALE / gym wrappers (environment.py:35-144) are emulator I/O and out of scope
(SURVEY.md §2.1); SyntheticEnvironment offers the same six methods on seeded uint8 frames so the
Agent loop and the benchmarks run without an emulator.
"""
import numpy as np


class Environment:
    def numActions(self):
        raise NotImplementedError

    def restart(self):
        raise NotImplementedError

    def act(self, action):
        raise NotImplementedError

    def getScreen(self):
        raise NotImplementedError

    def isTerminal(self):
        raise NotImplementedError

    def setMode(self, mode):
        pass


class SyntheticEnvironment(Environment):
    def __init__(self, args=None, num_actions=4, seed=0, terminal_prob=0.005, screen_height=84, screen_width=84):
        h = getattr(args, "screen_height", screen_height)
        w = getattr(args, "screen_width", screen_width)
        self.dims = (h, w)
        self._n = num_actions
        self._rng = np.random.RandomState(seed)
        self._tp = terminal_prob
        self._terminal = False
        self._screen = np.zeros(self.dims, dtype=np.uint8)
        self.mode = "train"

    def numActions(self):
        return self._n

    def restart(self):
        self._terminal = False
        self._screen = self._rng.randint(0, 256, size=self.dims, dtype=np.uint8)

    def act(self, action):
        assert 0 <= action < self._n
        self._screen = self._rng.randint(0, 256, size=self.dims, dtype=np.uint8)
        self._terminal = bool(self._rng.rand() < self._tp)
        return int(self._rng.randint(-1, 2))

    def getScreen(self):
        return self._screen

    def isTerminal(self):
        return self._terminal

    def setMode(self, mode):
        self.mode = mode
