#!/usr/bin/env python3
"""GAE + swap_and_flatten micro-benchmark (T = 600, N = 65 536): us and fraction of the 8 TB/s peak per kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
N, T = 65536, 600
env = qa.VecDockingEnv("docking-v0", num_envs=N)
def timed(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
g = torch.Generator(device="cuda").manual_seed(0)
rew = torch.randn((T, N), device="cuda", generator=g); val = torch.randn((T, N), device="cuda", generator=g)
dn8 = (torch.rand((T, N), device="cuda", generator=g) < 0.02).to(torch.uint8)
lv = torch.randn(N, device="cuda", generator=g); ld8 = (torch.rand(N, device="cuda", generator=g) < 0.1).to(torch.uint8)
us = timed(lambda: qa.compute_gae(env, rew, val, dn8, lv, ld8, 0.99, 0.95))
res = ["gae %.0f us %.2f" % (us, T * N * 17 / us / 1e3 / 8000)]
for d in (1, 4, 12):
    x = torch.randn((T, N, d) if d > 1 else (T, N), device="cuda", generator=g)
    us = timed(lambda: qa.swap_and_flatten(env, x), 10, 2)
    res.append("flatten d=%d %.0f us %.2f" % (d, us, 2 * x.numel() * 4 / us / 1e3 / 8000))
print(os.environ.get("QUADSIM_HIP_LIB", "default").split("/")[-1], " | ".join(res))
