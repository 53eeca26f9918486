"""EXPERIMENT: the fence-less step chain split over Q private queues (Q groups of tiles, one packet per queue and step)."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import quadsim_amd as qa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K, PER = 2048, 16
lib = qa._lib.load()
x = C.CDLL(os.path.join(R, "tools", "libqs_hsa_exp.so"))
x.qsx_error.restype = C.c_char_p
x.qsx_run_chain_multi.argtypes = [C.c_uint32] * 5 + [C.c_int, C.c_int, C.POINTER(C.c_double)]
def ck(rc, what):
    if rc != 0: raise RuntimeError("%s: %s" % (what, x.qsx_error().decode()))
def full_state(env):
    st = env.get_state()
    return np.concatenate([st["chaser"], st["target"], st["u_prev"], st["qdes"], st["last_shaping"][:, None], st["t"][:, None]], 1)
kw = dict(num_envs=n, randomise=1, seed=1234, init_range=qa.C3_INIT_RANGE, copy=False)
env, twin = qa.VecDockingEnv("docking-v0", **kw), qa.VecDockingEnv("docking-v0", **kw)
pool = env.random_actions(PER, step0=0)
p = lambda t: C.c_void_p(t.data_ptr())
ck(x.qsx_open(os.path.join(R, "tools", "quadsim_dev.hsaco").encode(), b"_ZN12_GLOBAL__N_111k_env_splitILi0ELb0ELi1EEEvNS_8StepArgsE.kd", 0, 8 * PER), "open")
lib.qs_debug_step_kernargs.argtypes = [C.c_void_p] * 8 + [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int64, C.c_int64]
buf = (C.c_char * 1024)(); size, split, tiles = C.c_uint64(), C.c_int32(), C.c_int64()
T = n // 64
for Q in (1, 2, 4, 8):
    per_q = T // Q
    for q in range(Q):
        for i in range(PER):
            assert lib.qs_debug_step_kernargs(env._h, p(pool[i]), p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term), buf, 1024,
                                              C.byref(size), C.byref(split), C.byref(tiles), q * per_q, (q + 1) * per_q) == 0
            ck(x.qsx_set_kernarg(q * PER + i, buf, int(size.value)), "kernarg")
    for acq, rel in ((1, 1), (1, 0)):
        env.reset(); twin.reset(); torch.cuda.synchronize()
        el = C.c_double()
        ck(x.qsx_run_chain_multi(K, Q, PER, per_q * 128, 128, acq, rel, C.byref(el)), "run")
        for k in range(K):
            twin.step(pool[k % PER])
        torch.cuda.synchronize()
        bad = int((full_state(env) != full_state(twin)).any(axis=1).sum())
        print("%d queue(s), release %-5s: %.2f us per step (%.2f G env-steps/s); envs differing from the HIP twin: %d"
              % (Q, "agent" if rel else "none", el.value / K, n * K / el.value / 1e3, bad))
x.qsx_close()
