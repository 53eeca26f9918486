#!/usr/bin/env python3
"""N sweep / mode sweep of the step kernel (GPU box): launch period per step and env-steps/s."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quadsim_amd import VecDockingEnv, C3_INIT_RANGE

def run(n, randomise=1, integ="frozen", K=1000, W=100, env_id="docking-v0", rollout_T=0):
    env = VecDockingEnv(env_id, num_envs=n, integrator=integ, randomise=randomise, seed=1, init_range=C3_INIT_RANGE,
                        mass_scale=(0.8, 1.2), inertia_scale=(0.8, 1.2))
    env.reset()
    P = 64
    pool = env.random_actions(P)
    if rollout_T:
        acts = pool[:abs(rollout_T)]
        sw = rollout_T < 0
        rollout_T = abs(rollout_T)
        out = env.rollout(acts, want_flags=True, stepwise=sw); torch.cuda.synchronize()
        reps = max(2, K // rollout_T)
        env.timer_start()
        for _ in range(reps):
            env.rollout(acts, stepwise=sw, out=out)
        ms = env.timer_stop()
        env.close()
        return ms * 1e3 / (reps * rollout_T), n * reps * rollout_T / (ms * 1e-3)
    lib, h = env._lib, env._h
    aptr = [C.c_void_p(pool[i].data_ptr()) for i in range(P)]
    args = (env._ptr(env._obs), env._ptr(env._rew), env._ptr(env._done), env._ptr(env._flags), env._ptr(env._term))
    for k in range(W):
        lib.qs_step(h, aptr[k % P], *args)
    torch.cuda.synchronize()
    env.timer_start()
    for k in range(K):
        lib.qs_step(h, aptr[k % P], *args)
    ms = env.timer_stop()
    env.close()
    return ms * 1e3 / K, n * K / (ms * 1e-3)

if __name__ == "__main__":
    print("lib:", os.environ.get("QUADSIM_HIP_LIB", "default"))
    for n in (4096, 16384, 65536, 131072, 262144, 524288, 1048576, 4194304):
        us, eps = run(n, K=1000 if n <= 262144 else 200)
        print("step  N=%8d  %8.2f us/step  %7.3f G env-steps/s  %6.1f GB/s algorithmic" % (n, us, eps / 1e9, eps * 392 / 1e9))
    for n in (4096, 65536, 131072, 262144):
        us, eps = run(n, rollout_T=-64, K=1024)
        print("stepwise(native loop) N=%8d  %8.2f us/step  %7.3f G env-steps/s" % (n, us, eps / 1e9))
    for r in (0, 1, 2):
        us, eps = run(65536, randomise=r)
        print("step  N=65536 randomise=%d  %8.2f us/step  %7.3f G/s" % (r, us, eps / 1e9))
    us, eps = run(65536, integ="rk4"); print("step  N=65536 rk4  %8.2f us  %7.3f G/s" % (us, eps / 1e9))
    us, eps = run(65536, env_id="docking-v2"); print("step  N=65536 v2  %8.2f us  %7.3f G/s" % (us, eps / 1e9))
    for n in (65536, 262144, 1048576):
        for T in (16, 64):
            us, eps = run(n, rollout_T=T, K=256)
            print("rollout N=%8d T=%3d  %8.2f us/step  %7.3f G env-steps/s" % (n, T, us, eps / 1e9))
