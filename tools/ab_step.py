#!/usr/bin/env python3
"""A/B of step-kernel builds on the raw qs_step loop (what bench.py times): us per step on the HIP stream and on 1 / 2 private
queues (host-ordered, pre-staged actions), with a small (cache-resident) and the headline's 512-batch action pool.
    QUADSIM_HIP_LIB=<twin>.so python tools/ab_step.py [--envs 65536] [--env docking-v0] [--randomise 1]
Prints one line per configuration; run the twins alternately (tools/ab_libs.sh) -- box-to-box and run-to-run spread is ~2 %."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import quadsim_amd as qa

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--env", default="docking-v0")
ap.add_argument("--randomise", type=int, default=1)
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--pools", default="16,512")
ap.add_argument("--modes", default="hip,q1,q2")
args = ap.parse_args()
n, K = args.envs, args.steps
res = {"lib": os.path.basename(os.environ.get("QUADSIM_HIP_LIB", "libquadsim_hip.so")), "envs": n}
for mode in args.modes.split(","):
    env = qa.VecDockingEnv(args.env, num_envs=n, randomise=args.randomise, seed=1234, init_range=qa.C3_INIT_RANGE,
                           mass_scale=(0.8, 1.2), inertia_scale=(0.8, 1.2), copy=False)
    env.reset()
    if mode != "hip":
        env.set_queue_mode(True, int(mode[1:]), ordering="host")
    lib, h = env._lib, env._h
    p = lambda t: C.c_void_p(t.data_ptr())            # noqa: E731
    io = (p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term))
    for P in [int(x) for x in args.pools.split(",")]:
        P = min(P, (1 << 29) // (n * 16))
        pool = env.random_actions(P, step0=0)
        ap_ = [p(pool[i]) for i in range(P)]
        seq = [ap_[k % P] for k in range(K)]
        best = 1e9
        for rep in range(3):
            for a in seq[:200]:
                lib.qs_step(h, a, *io)
            torch.cuda.synchronize(); env.sync()
            t0 = time.perf_counter()
            for a in seq:
                lib.qs_step(h, a, *io)
            env.sync(); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K * 1e6)
        res["%s/pool%d" % (mode, P)] = round(best, 3)
        del pool
    env.close()
print(json.dumps(res))
