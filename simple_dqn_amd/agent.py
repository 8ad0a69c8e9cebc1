"""Agent — py3 restatement of /root/reference/src/agent.py:7-135 (the caller of the hot path).

Logic is the reference's, line for line in behaviour (py2-only syntax replaced: `<>` :70, xrange).
The one addition: when the replay memory is device-backed (ours), Agent.train's inner
{getMinibatch ; net.train} pair (agent.py:110-114) runs as net.train_from_memory(mem), which
samples the same indexes from the same global random stream but never materialises the minibatch
on the host.  Set fused=False to force the reference's two-call form.
"""
import logging
import random

import numpy as np

from .state_buffer import DeviceStateBuffer, StateBuffer

logger = logging.getLogger(__name__)


class Agent:
    def __init__(self, environment, replay_memory, deep_q_network, args, fused=True):
        self.env = environment
        self.mem = replay_memory
        self.net = deep_q_network
        # acting state resident on the device when the network can read it there
        self.buf = DeviceStateBuffer(args) if hasattr(deep_q_network, "predict_state") else StateBuffer(args)
        self.num_actions = self.env.numActions()
        self.random_starts = args.random_starts
        self.history_length = args.history_length

        self.exploration_rate_start = args.exploration_rate_start
        self.exploration_rate_end = args.exploration_rate_end
        self.exploration_decay_steps = args.exploration_decay_steps
        self.exploration_rate_test = args.exploration_rate_test
        self.total_train_steps = args.start_epoch * args.train_steps

        self.train_frequency = args.train_frequency
        self.train_repeat = args.train_repeat
        self.target_steps = args.target_steps

        self.callback = None
        self.fused = fused and hasattr(self.net, "train_from_memory") and hasattr(self.mem, "_h")

    def _restartRandom(self):                                       # agent.py:29-39
        self.env.restart()
        for i in range(random.randint(self.history_length, self.random_starts) + 1):
            reward = self.env.act(0)
            terminal = self.env.isTerminal()
            if terminal:
                self.env.restart()
            screen = self.env.getScreen()
            self.buf.add(screen)

    def _explorationRate(self):                                     # :41-46
        if self.total_train_steps < self.exploration_decay_steps:
            return self.exploration_rate_start - self.total_train_steps * \
                (self.exploration_rate_start - self.exploration_rate_end) / self.exploration_decay_steps
        else:
            return self.exploration_rate_end

    def step(self, exploration_rate):                               # :48-85
        if random.random() < exploration_rate:
            action = random.randrange(self.num_actions)
        else:
            if hasattr(self.buf, "_h") and hasattr(self.net, "predict_state"):
                q0 = self.net.predict_state(self.buf)               # state already in HBM: one 7 KB upload per env step
            elif hasattr(self.net, "predict_one"):
                # batch-1 fast path: same numbers as predict(getStateMinibatch())[0], no zero-row padding
                q0 = self.net.predict_one(self.buf.getState())
            else:
                state = self.buf.getStateMinibatch()
                qvalues = self.net.predict(state)
                q0 = qvalues[0]
            assert len(q0) == self.num_actions
            action = int(np.argmax(q0))
        reward = self.env.act(action)
        screen = self.env.getScreen()
        terminal = self.env.isTerminal()
        self.buf.add(screen)
        if terminal:
            self._restartRandom()
        if self.callback:
            self.callback.on_step(action, reward, terminal, screen, exploration_rate)
        return action, reward, screen, terminal

    def play_random(self, random_steps):                            # :87-94
        self.env.restart()
        for i in range(random_steps):
            action, reward, screen, terminal = self.step(1)
            self.mem.add(action, reward, screen, terminal)

    def train(self, train_steps, epoch=0):                          # :96-116
        for i in range(train_steps):
            action, reward, screen, terminal = self.step(self._explorationRate())
            self.mem.add(action, reward, screen, terminal)
            if self.target_steps and i % self.target_steps == 0:
                self.net.update_target_network()
            if self.mem.count > self.mem.batch_size and i % self.train_frequency == 0:
                if self.fused:
                    if hasattr(self.net, "set_epoch"):
                        self.net.set_epoch(epoch)                   # net.train(minibatch, epoch), agent.py:114
                    self.net.train_from_memory(self.mem, self.train_repeat)
                else:
                    for j in range(self.train_repeat):
                        minibatch = self.mem.getMinibatch()
                        self.net.train(minibatch, epoch)
            self.total_train_steps += 1

    def test(self, test_steps, epoch=0):                            # :118-124
        self._restartRandom()
        for i in range(test_steps):
            self.step(self.exploration_rate_test)

    def play(self, num_games):                                      # :126-135
        self._restartRandom()
        for i in range(num_games):
            terminal = False
            while not terminal:
                action, reward, screen, terminal = self.step(self.exploration_rate_test)
                self.mem.add(action, reward, screen, terminal)
