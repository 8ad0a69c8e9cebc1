"""python -m simple_dqn_amd.main — the reference's top-level loop (/root/reference/src/main.py:16-165) on the
MI355X hot path.  Same flags and defaults; `--environment synthetic` (default here) replaces the ALE / gym
wrappers, which are emulator I/O outside the hot path.  Plumbing for BASELINE.json configs[0]."""
import argparse
import logging
import random
import sys


def _flag(text):
    return text.lower() in ("yes", "true", "t", "1")


# The command line of the reference (same flag names, defaults and grouping: src/main.py:16-84), as data.
# (flag, type, default[, extra argparse keywords])
_OPTIONS = {
    "Environment": [
        ("--environment", str, "synthetic", dict(choices=["synthetic", "ale", "gym"])),
        ("--num_actions", int, 4, dict(help="Action-set size of the synthetic environment.")),
        ("--synthetic_frame_pool", int, 256, dict(help="Synthetic environment: serve frames from a pool of this many pre-generated frames (0: generate 7 KB of random bytes every step).")),
        ("--screen_width", int, 84), ("--screen_height", int, 84),
    ],
    "Replay memory": [("--replay_size", int, 1000000), ("--history_length", int, 4)],
    "Deep Q-learning network": [
        ("--learning_rate", float, 0.00025), ("--discount_rate", float, 0.99), ("--batch_size", int, 32),
        ("--optimizer", str, "rmsprop", dict(choices=["rmsprop", "adam", "adadelta"])),
        ("--decay_rate", float, 0.95), ("--clip_error", float, 1), ("--min_reward", float, -1), ("--max_reward", float, 1),
        ("--batch_norm", _flag, False),
    ],
    "Backend": [
        ("--backend", str, "hip", dict(choices=["hip", "gpu", "cpu"])), ("--device_id", int, 0),
        ("--datatype", str, "float32", dict(choices=["float16", "float32", "float64"])),
        ("--stochastic_round", int, False, dict(const=True, nargs="?")),
    ],
    "Agent": [
        ("--exploration_rate_start", float, 1), ("--exploration_rate_end", float, 0.1),
        ("--exploration_decay_steps", float, 1000000), ("--exploration_rate_test", float, 0.05),
        ("--train_frequency", int, 4), ("--train_repeat", int, 1), ("--target_steps", int, 10000), ("--random_starts", int, 30),
    ],
    "Main loop": [
        ("--random_steps", int, 50000), ("--train_steps", int, 250000), ("--test_steps", int, 125000), ("--epochs", int, 200),
        ("--start_epoch", int, 0), ("--play_games", int, 0),
        ("--load_weights", str, None), ("--save_weights_prefix", str, None), ("--csv_file", str, None),
    ],
    "Common": [
        ("--random_seed", int, None),
        ("--log_level", str, "INFO", dict(choices=["DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"])),
    ],
}


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    for title, options in _OPTIONS.items():
        group = parser.add_argument_group(title)
        if title == "Environment":
            group.add_argument("game", nargs="?", default="synthetic", help="gym environment id (ignored by the synthetic environment)")
        for flag, typ, default, *extra in options:
            group.add_argument(flag, type=typ, default=default, **(extra[0] if extra else {}))
    return parser


def run(args):
    from . import Agent, DeepQNetwork, ReplayMemory, SyntheticEnvironment, _lib, load
    from .statistics import Statistics
    logger = logging.getLogger()
    logger.setLevel(args.log_level)
    if args.random_seed:                                             # main.py:89-90
        random.seed(args.random_seed)
    if args.device_id:
        _lib.check(load().sdqn_set_device(args.device_id))
    if args.environment == "ale":
        raise NotImplementedError("the ALE wrapper (src/environment.py:35-110) is emulator I/O outside the hot path; "
                                  "use --environment gym with gymnasium[atari] installed, or the synthetic environment")
    if args.environment == "gym":
        from .environment import GymEnvironment                    # needs gymnasium (or gym); not part of this image
        env = GymEnvironment(args.game, args)
    else:
        env = SyntheticEnvironment(args, num_actions=args.num_actions, seed=args.random_seed or 0, frame_pool=args.synthetic_frame_pool)
    mem = ReplayMemory(args.replay_size, args)                       # main.py:103-106
    net = DeepQNetwork(env.numActions(), args)
    agent = Agent(env, mem, net, args)
    stats = Statistics(agent, net, mem, env, args)
    if args.load_weights:
        net.load_weights(args.load_weights)
    if args.play_games:                                              # :112-128 (visualisation out of scope)
        env.setMode('test')
        stats.reset()
        agent.play(args.play_games)
        stats.write(0, "play")
        return stats
    if args.random_steps:                                            # :130-137
        env.setMode('train')
        stats.reset()
        agent.play_random(args.random_steps)
        stats.write(0, "random")
    for epoch in range(args.start_epoch, args.epochs):               # :140-162
        if args.train_steps:
            env.setMode('train')
            stats.reset()
            agent.train(args.train_steps, epoch)
            stats.write(epoch + 1, "train")
            if args.save_weights_prefix:
                net.save_weights(args.save_weights_prefix + "_%d.npz" % (epoch + 1))
        if args.test_steps:
            env.setMode('test')
            stats.reset()
            agent.test(args.test_steps, epoch)
            stats.write(epoch + 1, "test")
    stats.close()
    return stats


if __name__ == "__main__":
    logging.basicConfig(format='%(asctime)s %(message)s')
    run(build_parser().parse_args())
    sys.exit(0)
