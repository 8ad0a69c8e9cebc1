"""StateBuffer — py3 restatement of /root/reference/src/state_buffer.py:3-27 (host numpy; feeds predict)."""
import numpy as np


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
