#!/usr/bin/env python3
"""GPU-bound rate of the step chain when the host is out of the way: blocks of B steps, each ONE native qs_rollout_stepwise
call (B packets per queue), outputs into a reused [B,N,...] ring, actions from a P-batch pool.  Compare with tools/ab_step.py
(one ctypes call per step)."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import quadsim_amd as qa
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--B", type=int, default=20)
ap.add_argument("--blocks", type=int, default=150)
ap.add_argument("--pool", type=int, default=500)
args = ap.parse_args()
n, B = args.envs, args.B
res = {"envs": n, "B": B, "pool": args.pool}
for mode in ("hip", "q1/host", "q2/host", "q3/host", "q1/stream", "q2/stream"):
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=1234, init_range=qa.C3_INIT_RANGE, copy=False)
    env.reset()
    if mode != "hip":
        env.set_queue_mode(True, int(mode[1]), ordering=mode.split("/")[1])
    pool = env.random_actions(args.pool, step0=0)
    nb = args.pool // B
    lib, h = env._lib, env._h
    p = lambda t: C.c_void_p(t.data_ptr())            # noqa: E731
    obs = torch.empty((B, n, 12), device="cuda"); rew = torch.empty((B, n), device="cuda")
    done = torch.empty((B, n), dtype=torch.uint8, device="cuda"); flags = torch.empty((B, n), dtype=torch.uint8, device="cuda")
    ptrs = [p(pool[i * B:(i + 1) * B]) for i in range(nb)]
    io = (p(obs), p(rew), p(done), p(flags))
    best = 1e9
    for rep in range(3):
        for i in range(10):
            lib.qs_rollout_stepwise(h, B, ptrs[i % nb], *io)
        torch.cuda.synchronize(); env.sync()
        t0 = time.perf_counter()
        for i in range(args.blocks):
            lib.qs_rollout_stepwise(h, B, ptrs[i % nb], *io)
        t_issue = time.perf_counter() - t0
        env.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (args.blocks * B) * 1e6
        best = min(best, dt)
    res[mode] = {"us_per_step": round(best, 3), "host_issue_us_per_step": round(t_issue / (args.blocks * B) * 1e6, 3)}
    env.close()
print(json.dumps(res))
