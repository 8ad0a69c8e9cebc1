"""Agent — the caller of the hot path, with the observable behaviour of /root/reference/src/agent.py:7-135 (same
public methods, same order of `random` draws and environment calls, same callback events) and its own structure:

    _Epsilon            linear exploration schedule (agent.py:41-46)
    _greedy_action()    picks the cheapest Q-value path the network offers
    _advance()          one environment transition + bookkeeping (agent.py:48-85)
    _learn()            the {getMinibatch ; net.train} pair (agent.py:108-114) — fused into ONE library call,
                        net.train_from_memory(mem), when the replay memory is device-backed: same indexes from the
                        same global random stream, no host minibatch.  fused=False forces the reference's two calls.
"""
import logging
import random

import numpy as np

from .state_buffer import DeviceStateBuffer, StateBuffer

logger = logging.getLogger(__name__)


class _Epsilon:
    """exploration rate annealed linearly over `decay_steps` training steps, then constant"""

    def __init__(self, start, end, decay_steps):
        self.start, self.end, self.decay_steps = start, end, decay_steps

    def at(self, train_step):
        if train_step >= self.decay_steps:
            return self.end
        return self.start - train_step * (self.start - self.end) / self.decay_steps


class Agent:
    def __init__(self, environment, replay_memory, deep_q_network, args, fused=True):
        self.env, self.mem, self.net = environment, replay_memory, deep_q_network
        # acting state resident on the device when the network can read it there
        on_device = hasattr(deep_q_network, "predict_state")
        self.buf = DeviceStateBuffer(args) if on_device else StateBuffer(args)
        self.num_actions = self.env.numActions()
        self.history_length, self.random_starts = args.history_length, args.random_starts
        self._epsilon = _Epsilon(args.exploration_rate_start, args.exploration_rate_end, args.exploration_decay_steps)
        self.exploration_rate_test = args.exploration_rate_test
        self.total_train_steps = args.start_epoch * args.train_steps
        self.train_frequency, self.train_repeat, self.target_steps = args.train_frequency, args.train_repeat, args.target_steps
        self.callback = None
        self.fused = fused and hasattr(self.net, "train_from_memory") and hasattr(self.mem, "_h")
        # one library call per environment transition (state-buffer add [+ ring add] [+ the next acting forward enqueued ahead of its use])
        self._one_call = hasattr(self.net, "act_step") and hasattr(self.buf, "_h")

    # ---- acting ------------------------------------------------------------------------------------------
    def _greedy_action(self):
        if self._one_call and hasattr(self.net, "act_greedy"):
            a = self.net.act_greedy(self.buf)                         # predict_state + argmax inside the library
            assert 0 <= a < self.num_actions
            return a
        if hasattr(self.buf, "_h") and hasattr(self.net, "predict_state"):
            q = self.net.predict_state(self.buf)                      # state already in HBM: one 7 KB upload per env step
        elif hasattr(self.net, "predict_one"):
            q = self.net.predict_one(self.buf.getState())             # same numbers as predict(padded batch)[0]
        else:
            q = self.net.predict(self.buf.getStateMinibatch())[0]     # the reference's padded minibatch, agent.py:55-58
        assert len(q) == self.num_actions
        return int(np.argmax(q))

    def _fresh_episode(self):
        """restart and idle through a random number of no-op frames so episodes do not all start alike (agent.py:29-39)"""
        self.env.restart()
        for _ in range(random.randint(self.history_length, self.random_starts) + 1):
            self.env.act(0)
            if self.env.isTerminal():
                self.env.restart()
            self.buf.add(self.env.getScreen())

    def _advance(self, exploration_rate, store=False):
        explore = random.random() < exploration_rate                  # draw order matters: shared global stream
        action = random.randrange(self.num_actions) if explore else self._greedy_action()
        reward = self.env.act(action)
        screen, terminal = self.env.getScreen(), self.env.isTerminal()
        ring = store and hasattr(self.mem, "_h")
        if self._one_call:
            # the acting forward of the new state is started now when the next step will most likely want it (greedy with probability
            # 1 - exploration_rate) and the episode goes on; same Q-values as a forward started at the next step
            self.net.act_step(self.buf, self.mem if ring else None, screen, action, reward, terminal,
                              speculate=(exploration_rate < 0.5 and not terminal))
        else:
            self.buf.add(screen)
        if terminal:
            self._fresh_episode()
        if self.callback:
            self.callback.on_step(action, reward, terminal, screen, exploration_rate)
        if store and not (self._one_call and ring):
            self.mem.add(action, reward, screen, terminal)
        return action, reward, screen, terminal

    def _advance_and_store(self, exploration_rate):
        return self._advance(exploration_rate, store=True)

    # ---- learning ----------------------------------------------------------------------------------------
    def _learn(self, epoch):
        if self.fused:
            if hasattr(self.net, "set_epoch"):
                self.net.set_epoch(epoch)                             # what net.train(minibatch, epoch) would receive
            self.net.train_from_memory(self.mem, self.train_repeat)
            return
        for _ in range(self.train_repeat):
            self.net.train(self.mem.getMinibatch(), epoch)

    # ---- the reference's public surface -------------------------------------------------------------------
    def step(self, exploration_rate):
        return self._advance(exploration_rate)

    def _restartRandom(self):
        self._fresh_episode()

    def _explorationRate(self):
        return self._epsilon.at(self.total_train_steps)

    def play_random(self, random_steps):                              # fill the replay memory with uniform-random play
        self.env.restart()
        for _ in range(random_steps):
            self._advance_and_store(1)

    def train(self, train_steps, epoch=0):
        for i in range(train_steps):
            self._advance_and_store(self._epsilon.at(self.total_train_steps))
            if self.target_steps and i % self.target_steps == 0:      # also fires at i == 0 of every call (agent.py:105)
                self.net.update_target_network()
            if self.mem.count > self.mem.batch_size and i % self.train_frequency == 0:
                self._learn(epoch)
            self.total_train_steps += 1

    def test(self, test_steps, epoch=0):
        self._fresh_episode()
        for _ in range(test_steps):
            self._advance(self.exploration_rate_test)

    def play(self, num_games):
        self._fresh_episode()
        for _ in range(num_games):
            terminal = False
            while not terminal:
                terminal = self._advance_and_store(self.exploration_rate_test)[3]
