"""host time per qs_step call (enqueue only) vs the GPU-side step period, HIP-stream mode and private-queue mode"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
n, K = 65536, 3000
env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=1234, init_range=qa.C3_INIT_RANGE, copy=False)
env.reset()
pool = env.random_actions(64, step0=0)
p = lambda t: C.c_void_p(t.data_ptr())
lib, h = env._lib, env._h
args = (p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term))
aptr = [p(pool[i]) for i in range(64)]
for mode in ("hip", "private", "private-native-loop"):
    env.set_queue_mode(mode != "hip")
    for k in range(200):
        lib.qs_step(h, aptr[k % 64], *args)
    env.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "private-native-loop":
        T = 64
        big = pool
        o = torch.empty((T, n, 12), device="cuda"); r = torch.empty((T, n), device="cuda"); d = torch.empty((T, n), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K // T):
            lib.qs_rollout_stepwise(h, T, p(big), p(o), p(r), p(d), None)
    else:
        for k in range(K):
            lib.qs_step(h, aptr[k % 64], *args)
    t1 = time.perf_counter()
    env.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    kk = (K // 64) * 64 if mode == "private-native-loop" else K
    print("%-20s host enqueue %.2f us per step, with the final drain %.2f us per step" % (mode, (t1 - t0) / kk * 1e6, (t2 - t0) / kk * 1e6))
env.close()
