"""GPU parity of the replay half (bit-exact): HIP gather + native sampler vs the reference KATs and the oracle."""
import json
import os
import random

import numpy as np
import pytest

from oracle.replay_numpy import ReplayOracle, synthetic_fill
from util import crc, make_args

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    import simple_dqn_amd
    return simple_dqn_amd


@pytest.mark.parametrize("flags", [1, 2])            # HBM mirror, zero-copy
def test_getminibatch_matches_reference_kats(sd, golden_dir, flags):
    kats = json.load(open(os.path.join(golden_dir, "replay_kat.json")))["kats"]
    for k in kats:
        m = sd.ReplayMemory(k["size"], make_args(batch_size=k["B"]), flags=flags)
        synthetic_fill(m, k["fill_seed"], count=k["count"], current=k["current"])
        m.sync_mirror()
        random.seed(k["seed"])
        for call in k["calls"]:
            pre, act, rew, post, term = m.getMinibatch()
            assert m.last_indexes.tolist() == call["indexes"]
            assert (crc(pre), crc(post), crc(act), crc(rew), crc(term)) == \
                (call["crc_pre"], call["crc_post"], call["crc_actions"], call["crc_rewards"], call["crc_terminals"])
            assert pre.dtype == np.uint8 and act.dtype == np.uint8 and rew.dtype == np.int64 and term.dtype == np.bool_
            assert pre is m.prestates and post is m.poststates          # aliased buffers, like the reference
            assert np.array_equal(pre[:, 1:], post[:, :-1])
        assert crc(np.array(random.getstate()[1], dtype=np.uint32)) == k["mt_after_crc"]


def test_add_path_matches_oracle(sd):
    size, B = 300, 16
    a = make_args(batch_size=B)
    m, o = sd.ReplayMemory(size, a), ReplayOracle(size, batch_size=B)
    rng = np.random.RandomState(4)
    for i in range(size + 57):                      # wraps the ring
        scr = rng.randint(0, 256, (84, 84), dtype=np.uint8)
        act, rew, term = int(rng.randint(0, 4)), int(rng.randint(-3, 4)), bool(rng.rand() < 0.02)
        m.add(act, rew, scr, term)
        o.add(act, rew, scr, term)
        assert (m.count, m.current) == (o.count, o.current)
    assert np.array_equal(m.screens, o.screens) and np.array_equal(m.rewards, o.rewards)
    assert np.array_equal(m.terminals, o.terminals) and np.array_equal(m.actions, o.actions)
    for seed in (1, 2, 3):
        random.seed(seed)
        got = [x.copy() for x in m.getMinibatch()]
        random.seed(seed)
        exp = o.getMinibatch()
        for x, y in zip(got, exp):
            assert np.array_equal(x, y)
    for idx in (0, 1, 2, 3, 10, size - 1, -1):
        assert np.array_equal(m.getState(idx), o.getState(idx))
    with pytest.raises(AssertionError):
        m.add(0, 0, np.zeros((80, 80), np.uint8), False)           # replay_memory.py:27


def test_gather_properties_large(sd):
    """Size-independent properties at a ring that does not fit a toy test: every gathered state is the
    contiguous window ring[i-4:i] / ring[i-3:i+1] (checked through a checksum of checksums)."""
    size, B = 60000, 256
    m = sd.ReplayMemory(size, make_args(batch_size=B))
    rng = np.random.RandomState(1)
    block = rng.randint(0, 256, (1000, 84, 84), dtype=np.uint8)
    for i in range(size // 1000):
        m.screens[i * 1000:(i + 1) * 1000] = np.roll(block, i, axis=0) ^ np.uint8(i)
    m.actions[:] = rng.randint(0, 6, size)
    m.rewards[:] = rng.randint(-1, 2, size)
    m.terminals[:] = rng.rand(size) < 0.005
    m.count, m.current = size, size // 3
    m.sync_mirror()
    random.seed(77)
    pre, act, rew, post, term = m.getMinibatch()
    idx = m.last_indexes
    for k in range(0, B, 17):
        assert np.array_equal(pre[k], m.screens[idx[k] - 4:idx[k]]) and np.array_equal(post[k], m.screens[idx[k] - 3:idx[k] + 1])
    assert np.array_equal(act, m.actions[idx]) and np.array_equal(rew, m.rewards[idx]) and np.array_equal(term, m.terminals[idx])
    assert not any(m.terminals[i - 4:i].any() for i in idx)
    assert not any(i >= m.current and i - 4 < m.current for i in idx)
    ms = m.bench_gather(idx, iters=20)
    assert 0 < ms < 50


def test_full_size_ring_64bit_offsets(sd):
    """BASELINE.json's full ring (1 M frames = 7.06 GB, byte offsets beyond 2^32): states gathered around the 4 GiB
    boundary and at the very end of the ring are the contiguous windows of the host ring, and the fused
    gather + train step (ring offsets computed inside conv1 forward and conv1 wgrad) gives exactly the weights of
    the host-minibatch path fed with the same states."""
    from bench import fill_ring
    from oracle.dqn_numpy import xavier_weights
    size, B, A = 1000000, 32, 4
    args = make_args(batch_size=B)
    m = sd.ReplayMemory(size, args)
    fill_ring(m, 5, A)
    edge = (1 << 32) // 7056                                          # first frame whose byte offset needs 33 bits
    idx = np.array([4, 5, edge - 2, edge - 1, edge, edge + 1, edge + 2, edge + 3, edge + 4, edge + 5,
                    edge + 6, 750000, 900001, size - 1, size - 2, size - 17] +
                   list(np.random.RandomState(3).randint(edge, size, B - 16)), dtype=np.int64)
    m.terminals[:] = False                                            # every index is admissible (host master; metadata below)
    m.sync_mirror()
    pre, act, rew, post, term = m.gather(idx)
    for k in range(B):
        i = int(idx[k])
        assert np.array_equal(pre[k], m.screens[i - 4:i]) and np.array_equal(post[k], m.screens[i - 3:i + 1]), (k, i)
    assert np.array_equal(act, m.actions[idx]) and np.array_equal(rew, m.rewards[idx])
    ws = xavier_weights(A, 7)
    n1, n2 = sd.DeepQNetwork(A, args), sd.DeepQNetwork(A, args)
    for n in (n1, n2):
        n.set_weights(ws, 0); n.update_target_network()
    mb = (pre.copy(), act.copy(), rew.copy(), post.copy(), term.copy())
    for _ in range(2):
        n1.train_indexes(m, idx)                                      # ring offsets on the device
        n2.train(mb)                                                  # staged host minibatch
    for i in range(5):
        assert np.array_equal(n1.get_layer(i), n2.get_layer(i)), i


def test_ring_checkpoint_roundtrip(sd, tmp_path):
    """save() / load() of the replay ring (additive): a restored memory samples and gathers exactly like the original."""
    size, B = 500, 16
    args = make_args(batch_size=B)
    m = sd.ReplayMemory(size, args)
    rng = np.random.RandomState(4)
    for i in range(730):                                              # wraps: count = size, current = 230
        m.add(int(rng.randint(0, 4)), int(rng.randint(-1, 2)), rng.randint(0, 256, (84, 84), dtype=np.uint8), bool(rng.rand() < 0.02))
    p = str(tmp_path / "ring.bin")
    m.save(p)
    m2 = sd.ReplayMemory(size, args)
    m2.load(p)
    assert (m2.count, m2.current) == (m.count, m.current) == (500, 230)
    assert np.array_equal(m2.screens, m.screens) and np.array_equal(m2.rewards, m.rewards)
    assert np.array_equal(m2.actions, m.actions) and np.array_equal(m2.terminals, m.terminals)
    random.seed(9); a = [x.copy() for x in m.getMinibatch()]
    random.seed(9); b = [x.copy() for x in m2.getMinibatch()]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    with pytest.raises(AssertionError):
        sd.ReplayMemory(size + 1, args).load(p)
