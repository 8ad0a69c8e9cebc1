"""python -m simple_dqn_amd.main — the reference's top-level loop (/root/reference/src/main.py:16-165) on the
MI355X hot path.  Same flags and defaults; `--environment synthetic` (default here) replaces the ALE / gym
wrappers, which are emulator I/O outside the hot path.  Plumbing for BASELINE.json configs[0]."""
import argparse
import logging
import random
import sys


def str2bool(v):
    return v.lower() in ("yes", "true", "t", "1")


def build_parser():
    parser = argparse.ArgumentParser()
    envarg = parser.add_argument_group('Environment')
    envarg.add_argument("game", nargs="?", default="synthetic", help="Ignored by the synthetic environment.")
    envarg.add_argument("--environment", choices=["synthetic", "ale", "gym"], default="synthetic")
    envarg.add_argument("--num_actions", type=int, default=4, help="Action-set size of the synthetic environment.")
    envarg.add_argument("--screen_width", type=int, default=84)
    envarg.add_argument("--screen_height", type=int, default=84)
    memarg = parser.add_argument_group('Replay memory')
    memarg.add_argument("--replay_size", type=int, default=1000000)
    memarg.add_argument("--history_length", type=int, default=4)
    netarg = parser.add_argument_group('Deep Q-learning network')
    netarg.add_argument("--learning_rate", type=float, default=0.00025)
    netarg.add_argument("--discount_rate", type=float, default=0.99)
    netarg.add_argument("--batch_size", type=int, default=32)
    netarg.add_argument('--optimizer', choices=['rmsprop', 'adam', 'adadelta'], default='rmsprop')
    netarg.add_argument("--decay_rate", type=float, default=0.95)
    netarg.add_argument("--clip_error", type=float, default=1)
    netarg.add_argument("--min_reward", type=float, default=-1)
    netarg.add_argument("--max_reward", type=float, default=1)
    netarg.add_argument("--batch_norm", type=str2bool, default=False)
    neonarg = parser.add_argument_group('Backend')
    neonarg.add_argument('--backend', choices=['hip', 'gpu', 'cpu'], default='hip')
    neonarg.add_argument('--device_id', type=int, default=0)
    neonarg.add_argument('--datatype', choices=['float16', 'float32', 'float64'], default='float32')
    neonarg.add_argument('--stochastic_round', const=True, type=int, nargs='?', default=False)
    antarg = parser.add_argument_group('Agent')
    antarg.add_argument("--exploration_rate_start", type=float, default=1)
    antarg.add_argument("--exploration_rate_end", type=float, default=0.1)
    antarg.add_argument("--exploration_decay_steps", type=float, default=1000000)
    antarg.add_argument("--exploration_rate_test", type=float, default=0.05)
    antarg.add_argument("--train_frequency", type=int, default=4)
    antarg.add_argument("--train_repeat", type=int, default=1)
    antarg.add_argument("--target_steps", type=int, default=10000)
    antarg.add_argument("--random_starts", type=int, default=30)
    mainarg = parser.add_argument_group('Main loop')
    mainarg.add_argument("--random_steps", type=int, default=50000)
    mainarg.add_argument("--train_steps", type=int, default=250000)
    mainarg.add_argument("--test_steps", type=int, default=125000)
    mainarg.add_argument("--epochs", type=int, default=200)
    mainarg.add_argument("--start_epoch", type=int, default=0)
    mainarg.add_argument("--play_games", type=int, default=0)
    mainarg.add_argument("--load_weights")
    mainarg.add_argument("--save_weights_prefix")
    mainarg.add_argument("--csv_file")
    comarg = parser.add_argument_group('Common')
    comarg.add_argument("--random_seed", type=int)
    comarg.add_argument("--log_level", choices=["DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"], default="INFO")
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
        env = SyntheticEnvironment(args, num_actions=args.num_actions, seed=args.random_seed or 0)
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
