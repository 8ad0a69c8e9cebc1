import cProfile, pstats, sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_dqn_amd as sd
from util import make_args
args = make_args(batch_size=32, replay_size=50000, random_steps=2000, exploration_decay_steps=100000)
random.seed(1)
env = sd.SyntheticEnvironment(args, num_actions=4, seed=1)
mem = sd.ReplayMemory(args.replay_size, args); net = sd.DeepQNetwork(4, args); agent = sd.Agent(env, mem, net, args)
agent.play_random(2000)
pr = cProfile.Profile(); pr.enable(); agent.train(20000, 0); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
