"""CPU test of the Statistics restatement (reference statistics.py:8-124) with stand-in agent/net/mem objects."""
import csv

import numpy as np

from simple_dqn_amd.statistics import COLUMNS, Statistics
from util import make_args


class _Net:
    train_iterations = 0
    callback = None

    def predict(self, states):
        return np.tile(np.arange(4, dtype=np.float32), (states.shape[0], 1))


class _Mem:
    count, batch_size = 100, 8
    prestates = np.zeros((8, 4, 84, 84), np.uint8)

    def getMinibatch(self):
        return self.prestates, None, None, None, None


class _Agent:
    callback = None
    total_train_steps = 42


def test_statistics_csv_columns_and_running_means(tmp_path):
    p = str(tmp_path / "s.csv")
    agent, net, mem = _Agent(), _Net(), _Mem()
    st = Statistics(agent, net, mem, None, make_args(csv_file=p))
    assert agent.callback is st and net.callback is st
    st.reset()
    for i, (r, t) in enumerate([(1, False), (0, False), (2, True), (-1, False), (1, True)]):
        st.on_step(0, r, t, None, 0.5)
    assert (st.num_games, st.min_game_reward, st.max_game_reward) == (2, 0, 3) and st.average_reward == 1.5
    for i, c in enumerate([1.0, 3.0]):
        net.train_iterations = i + 1
        st.on_train(c)
    assert st.average_cost == 2.0                      # running mean over train_iterations (:71)
    st.write(1, "train")
    st.close()
    rows = list(csv.reader(open(p)))
    assert tuple(rows[0]) == COLUMNS and len(rows[0]) == 16
    assert rows[1][:4] == ["1", "train", "5", "2"] and float(rows[1][10]) == 3.0 and rows[1][8] == "42"
    assert st.validation_states is mem.prestates       # the reference's aliasing (:85-86)


def test_deferred_cost_protocol_keeps_the_running_mean_bit_identical(tmp_path):
    """on_train_deferred(collect, train_iterations): the cost is collected when the next step is announced, when average_cost is read, or
    when the phase ends — with the (cost, train_iterations) pairs and the order the immediate on_train would have used."""
    rng = np.random.RandomState(3)
    costs = rng.rand(50).astype(np.float32).tolist()
    a = Statistics(_Agent(), _Net(), _Mem(), None, make_args(csv_file=None))
    b = Statistics(_Agent(), _Net(), _Mem(), None, make_args(csv_file=None))
    collected = []
    for i, c in enumerate(costs):
        a.net.train_iterations = b.net.train_iterations = i + 1
        a.on_train(c)
        b.on_train_deferred(lambda c=c, i=i: (collected.append(i), c)[1], i + 1)
        assert len(collected) == i                     # only the PREVIOUS step's cost has been asked for
        if i == 20:
            assert b.average_cost == a.average_cost and len(collected) == 21      # reading it resolves what is pending
    b.net.train_iterations = 999                        # (a later value must not leak into the pending update)
    assert b.average_cost == a.average_cost and collected == list(range(50))
    b.on_train_deferred(lambda: 7.0, 51); a.net.train_iterations = 51; a.on_train(7.0)
    b.reset(); a.reset()                                # a phase end collects before the tally is replaced
    assert b.average_cost == a.average_cost == 0 and not b._pending


def test_write_collects_a_deferred_cost_that_is_still_outstanding(tmp_path):
    """the last train step's cost of a phase is still pending when main.py calls write(); the CSV's meancost column must hold
    the same running mean the immediate on_train form gives (4.0, 8.0 -> 6.0), not the mean without the last cost."""
    rows = {}
    for form in ("immediate", "deferred"):
        p = str(tmp_path / (form + ".csv"))
        net = _Net()
        st = Statistics(_Agent(), net, _Mem(), None, make_args(csv_file=p))
        for i, c in enumerate([4.0, 8.0]):
            net.train_iterations = i + 1
            if form == "immediate":
                st.on_train(c)
            else:
                st.on_train_deferred(lambda c=c: c, i + 1)
        st.write(1, "train")
        st.close()
        rows[form] = list(csv.reader(open(p)))[1]
    assert float(rows["deferred"][11]) == float(rows["immediate"][11]) == 6.0
