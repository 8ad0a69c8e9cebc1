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
    """Seeded uint8 frames, rewards in {-1, 0, 1}, terminal with probability terminal_prob.  frame_pool > 0 (default 256): the frames are
    drawn from a pool of that many frames generated once (an emulator's screen buffer costs nothing to read either; generating 7 KB of
    random bytes per step was 10 us of the 38 us an acting step took and said nothing about the library); frame_pool = 0 generates a new
    frame every step (the behaviour of rounds 1-3)."""

    def __init__(self, args=None, num_actions=4, seed=0, terminal_prob=0.005, screen_height=84, screen_width=84, frame_pool=256):
        h = getattr(args, "screen_height", screen_height)
        w = getattr(args, "screen_width", screen_width)
        self.dims = (h, w)
        self._n = num_actions
        self._rng = np.random.RandomState(seed)
        self._tp = terminal_prob
        self._terminal = False
        self._screen = np.zeros(self.dims, dtype=np.uint8)
        self._pool = self._rng.randint(0, 256, size=(frame_pool,) + self.dims, dtype=np.uint8) if frame_pool else None
        self.mode = "train"

    def _frame(self):
        if self._pool is None:
            return self._rng.randint(0, 256, size=self.dims, dtype=np.uint8)
        return self._pool[self._rng.randint(len(self._pool))]

    def numActions(self):
        return self._n

    def restart(self):
        self._terminal = False
        self._screen = self._frame()

    def act(self, action):
        assert 0 <= action < self._n
        self._screen = self._frame()
        self._terminal = bool(self._rng.rand() < self._tp)
        return int(self._rng.randint(-1, 2))

    def getScreen(self):
        return self._screen

    def isTerminal(self):
        return self._terminal

    def setMode(self, mode):
        self.mode = mode


def _to_gray_resized(obs, height, width):
    """cv2.resize(cv2.cvtColor(obs, COLOR_RGB2GRAY), (w, h)) of environment.py:139 without cv2: ITU-R 601 luma,
    then bilinear interpolation with half-pixel centres (cv2.INTER_LINEAR's sampling grid)."""
    obs = np.asarray(obs)
    if obs.ndim == 3:
        g = obs[..., 0] * 0.299 + obs[..., 1] * 0.587 + obs[..., 2] * 0.114
    else:
        g = obs.astype(np.float64)
    H, W = g.shape
    if (H, W) == (height, width):
        return np.clip(np.rint(g), 0, 255).astype(np.uint8)
    ys = (np.arange(height) + 0.5) * H / height - 0.5
    xs = (np.arange(width) + 0.5) * W / width - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, H - 1); y1 = np.clip(y0 + 1, 0, H - 1); fy = np.clip(ys - np.floor(ys), 0, 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, W - 1); x1 = np.clip(x0 + 1, 0, W - 1); fx = np.clip(xs - np.floor(xs), 0, 1)
    fy = np.where(ys < 0, 0.0, fy)[:, None]; fx = np.where(xs < 0, 0.0, fx)[None, :]
    top = g[y0][:, x0] * (1 - fx) + g[y0][:, x1] * fx
    bot = g[y1][:, x0] * (1 - fx) + g[y1][:, x1] * fx
    return np.clip(np.rint(top * (1 - fy) + bot * fy), 0, 255).astype(np.uint8)


class GymEnvironment(Environment):
    """py3 restatement of /root/reference/src/environment.py:112-144 for `gymnasium` (or classic `gym`) when one of
    them is installed — neither is in this image, so this adapter is exercised against a stand-in module only
    (tests/test_environment.py).  Emulator I/O is outside the hot path (optional part)."""

    def __init__(self, env_id, args, make=None):
        if make is None:
            try:
                import gymnasium as gym
            except ImportError:
                import gym
            make = gym.make
        self.gym = make(env_id)
        self.obs = None
        self.terminal = None
        self.screen_width = args.screen_width
        self.screen_height = args.screen_height
        self.mode = "train"

    def numActions(self):
        n = getattr(self.gym.action_space, "n", None)
        assert n is not None, "a Discrete action space is required"          # environment.py:127
        return int(n)

    def restart(self):
        r = self.gym.reset()
        self.obs = r[0] if isinstance(r, tuple) else r                         # gymnasium returns (obs, info)
        self.terminal = False

    def act(self, action):
        r = self.gym.step(action)
        if len(r) == 5:                                                       # gymnasium: terminated, truncated
            self.obs, reward, terminated, truncated, _ = r
            self.terminal = bool(terminated or truncated)
        else:                                                                 # classic gym, environment.py:135
            self.obs, reward, self.terminal, _ = r
            self.terminal = bool(self.terminal)
        return reward

    def getScreen(self):
        assert self.obs is not None
        return _to_gray_resized(self.obs, self.screen_height, self.screen_width)

    def isTerminal(self):
        assert self.terminal is not None
        return self.terminal

    def setMode(self, mode):
        self.mode = mode
