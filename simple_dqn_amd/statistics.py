"""Run statistics with the reference's observable behaviour (callbacks, log lines, 16-column CSV —
/root/reference/src/statistics.py:8-124), organised as: a per-phase tally (`_PhaseTally`), a CSV sink (`_CsvSink`)
and the `Statistics` facade that Agent / DeepQNetwork call back into.

Observability only; it is the second caller of mem.getMinibatch() (statistics.py:85) and net.predict() (:90).
Clock: CPU time (`time.process_time()`), which is what the reference's `time.clock()` returned on Linux.
Kept quirks: `validation_states` is whatever array getMinibatch() returned first (the reference aliases
mem.prestates, :85-86 — with the fused Agent path nothing overwrites it later; README.md), and the training-cost
average divides by the net's GLOBAL train_iterations although it restarts every phase (:55,71).
"""
import csv
import logging
import sys
import time

import numpy as np

logger = logging.getLogger(__name__)

COLUMNS = ("epoch", "phase", "steps", "nr_games", "average_reward", "min_game_reward", "max_game_reward",
           "last_exploration_rate", "total_train_steps", "replay_memory_count", "meanq", "meancost",
           "weight_updates", "total_time", "epoch_time", "steps_per_second")


class _PhaseTally:
    """Everything that restarts with reset(): step / game counters, reward extremes, running means."""

    def __init__(self):
        self.began = time.process_time()
        self.num_steps = self.num_games = 0
        self.game_rewards = self.average_reward = self.average_cost = 0
        self.min_game_reward, self.max_game_reward = sys.maxsize, -sys.maxsize - 1
        self.last_exploration_rate = 1

    def step(self, reward, terminal, exploration_rate):
        self.num_steps += 1
        self.game_rewards += reward
        self.last_exploration_rate = exploration_rate
        if not terminal:
            return
        self.num_games += 1                                              # a game ended: fold its return in
        self.average_reward += float(self.game_rewards - self.average_reward) / self.num_games
        self.min_game_reward = min(self.min_game_reward, self.game_rewards)
        self.max_game_reward = max(self.max_game_reward, self.game_rewards)
        self.game_rewards = 0

    def close_open_game(self):
        if self.num_games == 0:                                          # no finished game in this phase: report the open one
            self.num_games, self.average_reward = 1, self.game_rewards


class _CsvSink:
    def __init__(self, path):
        self.path = path
        self.fh = self.writer = None
        if path:
            logger.info("Results are written to %s" % path)
            self.fh = open(path, "w", newline="")
            self.writer = csv.writer(self.fh)
            self.row(COLUMNS)

    def row(self, values):
        if self.writer:
            self.writer.writerow(values)
            self.fh.flush()

    def close(self):
        if self.fh:
            self.fh.close()


_TALLY_FIELDS = ("num_steps", "num_games", "game_rewards", "average_reward", "min_game_reward", "max_game_reward",
                 "last_exploration_rate", "average_cost")


class Statistics:
    def __init__(self, agent, net, mem, env, args):
        self.agent, self.net, self.mem, self.env = agent, net, mem, env
        agent.callback = net.callback = self                            # both producers report here
        self.csv_name = args.csv_file
        self._csv = _CsvSink(self.csv_name)
        self.start_time = time.process_time()
        self.validation_states = None
        self.reset()

    # the tally's fields are part of the reference's public surface (st.num_games, st.average_cost, ...)
    def __getattr__(self, name):
        if name in _TALLY_FIELDS:
            return getattr(self.__dict__["_tally"], name)
        raise AttributeError(name)

    def reset(self):
        self._tally = _PhaseTally()
        self.epoch_start_time = self._tally.began

    def on_step(self, action, reward, terminal, screen, exploration_rate):
        self._tally.step(reward, terminal, exploration_rate)

    def on_train(self, cost):
        t = self._tally
        t.average_cost += (cost - t.average_cost) / self.net.train_iterations

    def _mean_max_q(self):
        if self.validation_states is None:
            return 0
        q = self.net.predict(self.validation_states)
        best = np.max(q, axis=1)
        assert best.shape[0] == q.shape[0]
        return np.mean(best)

    def write(self, epoch, phase):
        t = self._tally
        now = time.process_time()
        total_time, epoch_time = now - self.start_time, max(now - t.began, 1e-9)
        rate = t.num_steps / epoch_time
        t.close_open_game()
        if self.validation_states is None and self.mem.count > self.mem.batch_size:
            self.validation_states = self.mem.getMinibatch()[0]         # fixed held-out states for the meanq column
        if self.csv_name:
            self._csv.row((epoch, phase, t.num_steps, t.num_games, t.average_reward, t.min_game_reward, t.max_game_reward,
                           t.last_exploration_rate, self.agent.total_train_steps, self.mem.count, self._mean_max_q(),
                           t.average_cost, self.net.train_iterations, total_time, epoch_time, rate))
        logger.info("  num_games: %d, average_reward: %f, min_game_reward: %d, max_game_reward: %d" %
                    (t.num_games, t.average_reward, t.min_game_reward, t.max_game_reward))
        logger.info("  last_exploration_rate: %f, epoch_time: %ds, steps_per_second: %d" %
                    (t.last_exploration_rate, epoch_time, rate))

    def close(self):
        self._csv.close()
