"""host time per qs_step call (enqueue only) vs the step period with the final drain: HIP-stream mode and 1..3 private queues"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 3000
env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=1234, init_range=qa.C3_INIT_RANGE, copy=False)
env.reset()
pool = env.random_actions(64, step0=0)
p = lambda t: C.c_void_p(t.data_ptr())
lib, h = env._lib, env._h
args = (p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term))
aptr = [p(pool[i]) for i in range(64)]
for q in (0, 1, 2, 3):
    env.set_queue_mode(q > 0, max(q, 1))
    for k in range(200):
        lib.qs_step(h, aptr[k % 64], *args)
    env.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        lib.qs_step(h, aptr[k % 64], *args)
    t1 = time.perf_counter()
    env.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-18s host enqueue %.2f us per step, with the final drain %.2f us per step" % ("HIP stream" if q == 0 else "%d private queue(s)" % q, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
env.close()
