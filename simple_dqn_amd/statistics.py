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
            if name == "average_cost" and self.__dict__.get("_pending"):
                self._resolve_pending()
            return getattr(self.__dict__["_tally"], name)
        raise AttributeError(name)

    def reset(self):
        self._resolve_pending()                                         # (costs of the phase that ends belong to its tally)
        self._tally = _PhaseTally()
        self.epoch_start_time = self._tally.began

    def on_step(self, action, reward, terminal, screen, exploration_rate):
        self._tally.step(reward, terminal, exploration_rate)

    def on_train(self, cost):
        self._resolve_pending()
        t = self._tally
        t.average_cost += (cost - t.average_cost) / self.net.train_iterations

    # Deferred form of on_train (simple_dqn_amd.DeepQNetwork.train_from_memory offers it to callbacks that have this method): the train
    # step has been ENQUEUED, its cost is collected later — when the next one is announced, or when average_cost is read / the phase
    # ends.  The running mean is updated with exactly the (cost, train_iterations) pairs and in exactly the order the immediate form
    # would have used (statistics.py:70-71 of the reference), so the tally is bit-identical; the host just no longer waits 65 us for
    # the GPU after every fourth environment step.
    def on_train_deferred(self, collect, train_iterations):
        self._resolve_pending()
        self._pending.append((collect, train_iterations))

    def _resolve_pending(self):
        pend = self.__dict__.get("_pending")
        if pend is None:
            self._pending = []
            return
        while pend:
            collect, iters = pend.pop(0)
            t = self._tally
            t.average_cost += (collect() - t.average_cost) / iters

    def _mean_max_q(self):
        if self.validation_states is None:
            return 0
        q = self.net.predict(self.validation_states)
        best = np.max(q, axis=1)
        assert best.shape[0] == q.shape[0]
        return np.mean(best)

    def write(self, epoch, phase):
        self._resolve_pending()                                         # a deferred cost still outstanding belongs in this row's meancost
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
