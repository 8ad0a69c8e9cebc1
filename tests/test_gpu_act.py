"""The acting forward as ONE launch (simple_dqn_amd/csrc/sdqn_act.hip; agent.py:48-59 -> deepqnetwork.py:175-184 for a batch of one state):
per-XCC redundant conv chain handed over through the shared L2, ticket-claimed work, fc4 split over the whole chip, last-arriver head.
Checked against the five-launch forward it replaces (the train step's forward kernels at B = 1, themselves oracle-tested), against the
batched predict(), for determinism (the claim order of the work items differs from launch to launch; the sums must not), and for staleness
(states and weights change between calls: every launch must see the current ones)."""
import ctypes as C

import numpy as np
import pytest

from oracle.dqn_numpy import OracleDQN, xavier_weights
from util import make_args, random_minibatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    import simple_dqn_amd
    return simple_dqn_amd


def _net(sd, A, seed, B=32):
    net = sd.DeepQNetwork(A, make_args(batch_size=B))
    net.set_weights(xavier_weights(A, seed), 0)
    net.update_target_network()
    return net


@pytest.mark.parametrize("A", [3, 4, 6, 18])
def test_one_launch_forward_matches_the_oracle_and_the_five_launch_forward(sd, A):
    net = _net(sd, A, 500 + A)
    orc = OracleDQN(A, batch_size=1, weights=xavier_weights(A, 500 + A))
    rng = np.random.RandomState(A)
    for i in range(6):
        state = rng.randint(0, 256, (4, 84, 84), dtype=np.uint8)
        if i == 4:
            state[:] = 0                                  # all-zero state: Q = 0 exactly (no biases, deepqnetwork.py:83-91)
        if i == 5:
            state[:] = 255
        net.set_option("act_kernel", 1)
        q1 = net.predict_one(state)
        net.set_option("act_kernel", 0)
        q5 = net.predict_one(state)
        qo = orc.predict(state[None])[0]
        scale = max(1e-3, float(np.abs(qo).max()))
        assert np.abs(q1 - q5).max() <= 2e-6 * scale + 1e-7, (i, q1, q5)          # fp32 sums in two different orders
        assert np.abs(q1 - qo).max() <= 2e-5 * scale + 1e-6, (i, q1, qo)          # float32 oracle (numpy), its own order
        if i == 4:
            assert np.all(q1 == 0)


def test_one_launch_forward_is_deterministic_and_never_stale(sd):
    """300 forwards with the state changing every call and the weights every few calls (train steps in between): each equals the
    five-launch forward of the same state and weights to fp32 round-off; the same state + weights twice gives the same bits (which
    workgroup computed which item differs between the two launches)."""
    A, B = 4, 32
    net = _net(sd, A, 601)
    buf = sd.DeviceStateBuffer(make_args(batch_size=B))
    rng = np.random.RandomState(602)
    for i in range(300):
        buf.add(rng.randint(0, 256, (84, 84), dtype=np.uint8))
        if i % 9 == 4:
            net.train(random_minibatch(B, A, 700 + i))
        net.set_option("act_kernel", 1)
        qa = net.predict_state(buf)
        qb = net.predict_state(buf)
        net.set_option("act_kernel", 0)
        q5 = net.predict_state(buf)
        assert np.array_equal(qa, qb), i
        assert np.abs(qa - q5).max() <= 2e-6 * max(1e-3, float(np.abs(q5).max())) + 1e-7, (i, qa, q5)
        assert np.array_equal(qa, net_predict_one_on(net, buf)), i


def net_predict_one_on(net, buf):
    net.set_option("act_kernel", 1)
    return net.predict_one(buf.getState())


def test_phase_stamps_hook(sd):
    """sdqn_net_debug_act: the same Q-values as the product launch, and per-workgroup {kind, clock} stamps that cover all 65 items of
    every XCC that received workgroups and all 256 fc4 items exactly once."""
    A = 6
    net = _net(sd, A, 611)
    buf = sd.DeviceStateBuffer(make_args(batch_size=32))
    rng = np.random.RandomState(612)
    for _ in range(5):
        buf.add(rng.randint(0, 256, (84, 84), dtype=np.uint8))
    q = np.empty(A, np.float32)
    st = np.zeros((256, 80), np.uint64)
    lib = sd.load()
    from simple_dqn_amd import _lib
    _lib.check(lib.sdqn_net_debug_act(net._h, buf._h, q.ctypes.data_as(C.POINTER(C.c_float)), st.ctypes.data_as(C.POINTER(C.c_uint64))))
    assert np.array_equal(q, net.predict_state(buf))
    kinds = st[:, 0:78:2].astype(np.int64)
    code, item = kinds >> 16, kinds & 0xFFFF
    xcc = st[:, 79].astype(np.int64)
    assert sorted(item[code == 7].tolist()) == list(range(256)) and sorted(item[code == 9].tolist()) == list(range(256))
    assert sorted(item[code == 10].tolist()) == list(range(8))            # every stripe's partial Q-vector written exactly once
    for x in set(xcc.tolist()):
        rows = xcc == x
        for c, n in ((1, 25), (3, 24), (5, 16)):
            assert sorted(item[rows][code[rows] == c].tolist()) == list(range(n)), (x, c)


def test_deferred_cost_gives_the_same_statistics_and_random_stream(sd):
    """Agent.train with the Statistics callback: by default the train step is only ENQUEUED and Statistics collects its cost later
    (on_train_deferred); a callback without that method gets the immediate on_train(cost) of deepqnetwork.py:168-172.  Same
    average_cost (bit for bit: same (cost, train_iterations) pairs in the same order), same weights, same ring, and Python's global
    generator — advanced by the number of words the library's sampler drew — ends in the same state."""
    import random
    A, B = 4, 32
    outs = []
    for deferred in (True, False):
        args = make_args(batch_size=B, replay_size=2000, random_steps=200, exploration_rate_start=0.5, exploration_rate_end=0.1,
                         exploration_decay_steps=300, target_steps=50, train_frequency=4)
        random.seed(args.random_seed)
        env = sd.SyntheticEnvironment(args, num_actions=A, seed=5)
        mem = sd.ReplayMemory(args.replay_size, args)
        net = sd.DeepQNetwork(A, args)
        net.set_weights(xavier_weights(A, 801), 0); net.update_target_network()
        agent = sd.Agent(env, mem, net, args)
        st = sd.Statistics(agent, net, mem, env, args)
        if not deferred:
            class Immediate:                                   # the reference's callback protocol only
                def on_step(self, *a): st.on_step(*a)
                def on_train(self, cost): st.on_train(cost)
            agent.callback = net.callback = Immediate()
        agent.play_random(args.random_steps)
        st.reset()
        agent.train(400, 0)
        costs = st.average_cost
        agent.test(50, 0)
        outs.append((costs, net.train_iterations, net.get_weights(0), random.getstate(), mem.count, mem.current))
    a, b = outs
    assert a[0] == b[0] and a[0] > 0 and a[1] == b[1] == 100 and a[3] == b[3] and a[4:] == b[4:]
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)


def test_a_launch_that_delivers_nothing_falls_back_to_the_five_launch_forward(sd, capfd):
    """Every poll inside the one-launch forward is bounded, and the host's wait for the Q partials is too (2 ms, then a stream
    synchronisation): a launch that delivers nothing (injected: its work appears claimed already) makes the library fall back to the
    five-launch forward for THIS call and every later one, with one line on stderr — same Q-values as act_kernel = 0."""
    A = 4
    net = _net(sd, A, 631)
    buf = sd.DeviceStateBuffer(make_args(batch_size=32))
    rng = np.random.RandomState(632)
    for _ in range(5):
        buf.add(rng.randint(0, 256, (84, 84), dtype=np.uint8))
    q_ok = net.predict_state(buf)
    net.set_option("act_inject_failure", 1)
    q_fb = net.predict_state(buf)                       # injected failure -> fallback inside this call
    q_after = net.predict_state(buf)                    # stays on the five launches
    net.set_option("act_kernel", 0)
    q5 = net.predict_state(buf)
    assert np.array_equal(q_fb, q5) and np.array_equal(q_after, q5)
    assert np.abs(q_ok - q5).max() <= 2e-6 * max(1e-3, float(np.abs(q5).max())) + 1e-7
    assert "did not complete" in capfd.readouterr().err
