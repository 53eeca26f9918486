"""EXPERIMENT: does the fence-less step chain (release scope NONE) depend on every tile being stepped by the SAME XCD in
consecutive launches?  Even steps: block b steps tile b.  Odd steps: block b steps tile b + 1 (another XCD: blocks are dealt
round-robin), tile 0 in a launch of its own.  If the state only lived in the stepping XCD's L2, the odd steps would read stale
state and the final state would differ from the HIP-launched twin."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import quadsim_amd as qa
n, K = 65536, 1024
lib = qa._lib.load()
x = C.CDLL(os.path.join(R, "tools", "libqs_hsa_exp.so"))
x.qsx_error.restype = C.c_char_p
x.qsx_run_chain.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_double)]
def ck(rc, what):
    if rc != 0: raise RuntimeError("%s: %s" % (what, x.qsx_error().decode()))
def full_state(env):
    st = env.get_state()
    return np.concatenate([st["chaser"], st["target"], st["u_prev"], st["qdes"], st["last_shaping"][:, None], st["t"][:, None]], 1)
kw = dict(num_envs=n, randomise=1, seed=1234, init_range=qa.C3_INIT_RANGE, copy=False)
env, twin = qa.VecDockingEnv("docking-v0", **kw), qa.VecDockingEnv("docking-v0", **kw)
pool = env.random_actions(2, step0=0)
p = lambda t: C.c_void_p(t.data_ptr())
ck(x.qsx_open(os.path.join(R, "tools", "quadsim_dev.hsaco").encode(), b"_ZN12_GLOBAL__N_111k_env_splitILi0ELb0ELi1EEEvNS_8StepArgsE.kd", 0, 8), "open")
lib.qs_debug_step_kernargs.argtypes = [C.c_void_p] * 8 + [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int64, C.c_int64]
buf = (C.c_char * 1024)(); size, split, tiles = C.c_uint64(), C.c_int32(), C.c_int64()
def slot(i, act, t0, t1):
    assert lib.qs_debug_step_kernargs(env._h, p(act), p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term), buf, 1024,
                                      C.byref(size), C.byref(split), C.byref(tiles), t0, t1) == 0
    ck(x.qsx_set_kernarg(i, buf, int(size.value)), "kernarg")
T = n // 64
slot(0, pool[0], 0, 0)            # even step: all tiles, block b -> tile b
slot(1, pool[1], 1, T)            # odd step, part 1: blocks 0..T-2 -> tiles 1..T-1 (shifted by one XCD)
slot(2, pool[1], 0, 1)            # odd step, part 2: tile 0
names = {0: "none", 1: "agent"}
for shifted in (False, True):
    for acq, rel in ((1, 1), (1, 0), (0, 0)):
        if shifted:
            pat_s = (C.c_uint32 * 3)(0, 1, 2); pat_g = (C.c_uint32 * 3)(T * 128, (T - 1) * 128, 128); npk = K // 2 * 3
            x.qsx_set_pattern(3, pat_s, pat_g)
        else:
            pat_s = (C.c_uint32 * 2)(0, 2 + 1); pat_g = (C.c_uint32 * 2)(T * 128, T * 128); npk = K
            slot(3, pool[1], 0, 0)
            x.qsx_set_pattern(2, pat_s, pat_g)
        env.reset(); twin.reset(); torch.cuda.synchronize()
        el = C.c_double()
        ck(x.qsx_run_chain(npk, T * 128, 128, acq, rel, C.byref(el)), "run")
        for k in range(K):
            twin.step(pool[k % 2])
        torch.cuda.synchronize()
        a, b = full_state(env), full_state(twin)
        bad = int((a != b).any(axis=1).sum())
        print("%-28s acquire %-5s release %-5s: %.2f us per step; envs whose final state differs from the HIP twin: %d of %d"
              % ("tiles change XCD every step" if shifted else "tiles stay on their XCD", names[acq], names[rel], el.value / K, bad, n))
x.qsx_close()
