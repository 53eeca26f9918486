#!/usr/bin/env python3
"""PCIe-inclusive step rate (never bench.py's `value`): numpy actions in, numpy obs / reward / done out, one H2D + D2H per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import quadsim_amd as qa
for n in (1, 4096, 65536):
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=0, init_range=qa.C3_INIT_RANGE, backend="numpy")
    env.reset()
    a = np.random.RandomState(0).uniform(-1, 1, (n, 4)).astype(np.float32)
    K = 300
    for _ in range(20):
        env.step(a)
    t0 = time.perf_counter()
    for _ in range(K):
        obs, rew, done, infos = env.step(a)
    dt = (time.perf_counter() - t0) / K
    print("backend=numpy N=%6d: %8.1f us/step  %10.3f M env-steps/s  (%.1f KB H2D + %.1f KB D2H per step)" % (
        n, dt * 1e6, n / dt / 1e6, n * 16 / 1e3, n * 53 / 1e3))
    env.close()
e = qa.DockingEnv(); e.reset()
t0 = time.perf_counter()
for _ in range(300):
    o, r, d, info = e.step(np.zeros(4))
    if d: e.reset()
print("single-env gym shim (QS_IO_HOST): %.1f us/step" % ((time.perf_counter() - t0) / 300 * 1e6))
