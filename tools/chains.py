#!/usr/bin/env python3
"""S independent env groups (handles with their own streams) of N/S envs each, stepped with native per-step launches:
does overlapping the groups' launch gaps / store drains raise total env-steps/s at fixed total N?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quadsim_amd import VecDockingEnv, C3_INIT_RANGE

def run(total, S, T=64, reps=16, fused=False):
    n = total // S
    envs = [VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=1, env_id_offset=i * n, init_range=C3_INIT_RANGE,
                          use_torch_stream=False) for i in range(S)]
    for e in envs: e.reset()
    acts = [e.random_actions(T) for e in envs]
    outs = [e.rollout(a, stepwise=not fused) for e, a in zip(envs, acts)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for e, a, o in zip(envs, acts, outs):
            e.rollout(a, stepwise=not fused, out=o)
    for e in envs: e.sync()
    dt = time.perf_counter() - t0
    for e in envs: e.close()
    return dt * 1e6 / (reps * T), total * reps * T / dt

for total in (65536, 262144):
    for S in (1, 2, 4, 8):
        us, eps = run(total, S)
        print("total N=%7d  groups=%d  %6.2f us per step of all groups  %7.3f G env-steps/s" % (total, S, us, eps / 1e9))
