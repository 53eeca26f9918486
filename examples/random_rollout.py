#!/usr/bin/env python3
"""Random-action roll-outs (the synthetic workload of BASELINE.json): the per-step VecEnv API, T single-step launches issued
natively (HIP stream, then the handle's private AQL queues: no end-of-kernel cache write-back, one GPU-side hand-shake with
torch's stream per call), and the fused T-step launch."""
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
# T single-step launches per call (every step's outputs kept), first on the HIP stream, then on two private queues: the call is
# ordered against torch's stream on the GPU (qs_set_queue_ordering: QS_ORDER_STREAM), so `S` below may be used right away
for private in (False, True):
    env.set_queue_mode(private, 2)
    bufs = env.rollout(actions, stepwise=True)       # warm-up / allocation
    torch.cuda.synchronize(); t0 = time.perf_counter()
    O2, R2, D2, F2 = env.rollout(actions, stepwise=True, out=bufs)
    S = R2.sum()                                      # consumed in stream order: no host synchronisation in between
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("stepwise roll-out, %-22s: %.2f G env-steps/s (reward sum %.3f)" % (
        "private queues (%s)" % env.queue_ordering if private else "HIP stream", N * T / (t1 - t0) / 1e9, float(S)))
env.set_queue_mode(False)
env.close()
