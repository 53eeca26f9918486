"""cProfile of VecDockingEnv.step (host side; small batch so that the GPU is never the limit)."""
import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
env = qa.VecDockingEnv("docking-v0", num_envs=1024, randomise=1, seed=1, init_range=qa.C3_INIT_RANGE, copy=("--copy" in sys.argv))
env.reset()
a = env.random_actions(1)[0]
for _ in range(500):
    env.step(a)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5000):
    env.step(a)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
