"""StateBuffer — py3 restatement of /root/reference/src/state_buffer.py:3-27 (host numpy; feeds predict), and
DeviceStateBuffer — the same interface with the last `history_length` screens resident in HBM
(sdqn_statebuf_*): add() uploads one 7 KB frame, DeepQNetwork.predict_state() reads it in place."""
import ctypes as C

import numpy as np

from . import _lib


class StateBuffer:
    def __init__(self, args):
        self.history_length = args.history_length
        self.dims = (args.screen_height, args.screen_width)
        self.batch_size = args.batch_size
        self.buffer = np.zeros((self.batch_size, self.history_length) + self.dims, dtype=np.uint8)

    def add(self, observation):
        assert observation.shape == self.dims
        self.buffer[0, :-1] = self.buffer[0, 1:]
        self.buffer[0, -1] = observation

    def getState(self):
        return self.buffer[0]

    def getStateMinibatch(self):
        return self.buffer

    def reset(self):
        self.buffer *= 0


class DeviceStateBuffer:
    """Drop-in for StateBuffer whose current state also lives on the device.  getState()/getStateMinibatch()
    return host copies with the reference's layout (row 0 = current state, other rows zero)."""

    def __init__(self, args):
        self.history_length = args.history_length
        self.dims = (args.screen_height, args.screen_width)
        self.batch_size = args.batch_size
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.sdqn_statebuf_create(C.byref(h), self.dims[0], self.dims[1], self.history_length))
        self._h = h
        self.buffer = np.zeros((self.batch_size, self.history_length) + self.dims, dtype=np.uint8)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and self._lib is not None:
            try:
                self._lib.sdqn_statebuf_destroy(h)
            except Exception:
                pass
            self._h = None

    def add(self, observation):                                     # state_buffer.py:15-18
        assert observation.shape == self.dims
        obs = np.ascontiguousarray(observation, dtype=np.uint8)
        _lib.check(self._lib.sdqn_statebuf_add(self._h, _lib.ptr(obs, C.c_uint8)))

    def getState(self):                                             # :20-21
        _lib.check(self._lib.sdqn_statebuf_get(self._h, _lib.ptr(self.buffer[0], C.c_uint8)))
        return self.buffer[0]

    def getStateMinibatch(self):                                    # :23-24
        self.getState()
        return self.buffer

    def reset(self):                                                # :26-27
        _lib.check(self._lib.sdqn_statebuf_reset(self._h))
        self.buffer *= 0
