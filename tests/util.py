"""Shared helpers for the tests (not product code)."""
import argparse
import zlib

import numpy as np


def make_args(**kw):
    """The argparse namespace of /root/reference/src/main.py:16-84 with its defaults."""
    d = dict(screen_width=84, screen_height=84, history_length=4, replay_size=1000000,
             learning_rate=0.00025, discount_rate=0.99, batch_size=32, optimizer="rmsprop", decay_rate=0.95,
             clip_error=1.0, min_reward=-1.0, max_reward=1.0, batch_norm=False, backend="hip", device_id=0,
             datatype="float32", stochastic_round=False, exploration_rate_start=1.0, exploration_rate_end=0.1,
             exploration_decay_steps=1000000, exploration_rate_test=0.05, train_frequency=4, train_repeat=1,
             target_steps=10000, random_starts=30, random_steps=50000, train_steps=250000, test_steps=125000,
             epochs=200, start_epoch=0, play_games=0, load_weights=None, save_weights_prefix=None, csv_file=None,
             random_seed=123, log_level="INFO")
    d.update(kw)
    return argparse.Namespace(**d)


def crc(a):
    return "%08x" % (zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)


def random_minibatch(B, A, seed, p_term=0.2, reward_range=(-2, 3)):
    rng = np.random.RandomState(seed)
    pre = rng.randint(0, 256, (B, 4, 84, 84), dtype=np.uint8)
    post = rng.randint(0, 256, (B, 4, 84, 84), dtype=np.uint8)
    act = rng.randint(0, A, B).astype(np.uint8)
    rew = rng.randint(reward_range[0], reward_range[1], B).astype(np.int64)
    term = rng.rand(B) < p_term
    return pre, act, rew, post, term
