#!/usr/bin/env python3
"""Random-action roll-outs (the synthetic workload of BASELINE.json): per-step API vs the fused T-step launch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import quadsim_amd as qa  # noqa: E402

N, T = 65536, 256
env = qa.VecDockingEnv("docking-v0", num_envs=N, randomise=1, seed=0, init_range=qa.C3_INIT_RANGE)
obs = env.reset()
actions = env.random_actions(T)                      # [T,N,4] U(-1,1) from the rocRAND action stream
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
    obs, rew, done, infos = env.step(actions[t])     # SB2 VecEnv protocol, torch tensors on the device
torch.cuda.synchronize(); t1 = time.perf_counter()
O, R, D, F = env.rollout(actions)                    # the same T steps in one launch
torch.cuda.synchronize(); t2 = time.perf_counter()
print("per-step API : %.2f G env-steps/s" % (N * T / (t1 - t0) / 1e9))
print("fused rollout: %.2f G env-steps/s   episodes ended %d, mean reward %.4f" % (
    N * T / (t2 - t1) / 1e9, int(D.sum()), float(R.mean())))
env.close()
