"""EXPERIMENT: the step kernel dispatched through an HSA queue of our own with explicit AQL fence scopes (tools/hsa_chain_exp.cpp).
Question: what do the agent-scope release + acquire that HIP attaches to every launch cost a chain of
dependent step launches, and does the chain stay CORRECT without them (tiles are re-read by the workgroup index that wrote them)?
    hipcc --cuda-device-only --offload-arch=gfx950 ... quadsim_hip.hip -o tools/quadsim_dev.hsaco   (tools/build_hsa_exp.sh)
    python tools/hsa_chain_exp.py [envs]
"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import quadsim_amd as qa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = 2048
lib = qa._lib.load()
x = C.CDLL(os.path.join(R, "tools", "libqs_hsa_exp.so"))
x.qsx_error.restype = C.c_char_p
x.qsx_run_chain.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_double)]


def ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s: %s" % (what, x.qsx_error().decode()))


def full_state(env):
    st = env.get_state()
    return np.concatenate([st["chaser"], st["target"], st["u_prev"], st["qdes"], st["last_shaping"][:, None], st["t"][:, None]], 1)


kw = dict(num_envs=n, randomise=1, seed=1234, init_range=qa.C3_INIT_RANGE, copy=False)
env, twin = qa.VecDockingEnv("docking-v0", **kw), qa.VecDockingEnv("docking-v0", **kw)
P = 64
pool = env.random_actions(P, step0=0)
p = lambda t: C.c_void_p(t.data_ptr())
sym = b"_ZN12_GLOBAL__N_111k_env_splitILi0ELb0ELi1EEEvNS_8StepArgsE.kd" if n <= 131072 else b"_ZN12_GLOBAL__N_15k_envILi0ELb0ELi1EEEvNS_8StepArgsE.kd"
ck(x.qsx_open(os.path.join(R, "tools", "quadsim_dev.hsaco").encode(), sym, 0, P), "qsx_open")
ks, gs, ps = C.c_uint32(), C.c_uint32(), C.c_uint32()
x.qsx_info(C.byref(ks), C.byref(gs), C.byref(ps))
print("kernel kernarg %d B, LDS %d B, scratch %d B" % (ks.value, gs.value, ps.value))
buf = (C.c_char * 1024)()
size, split, tiles = C.c_uint64(), C.c_int32(), C.c_int64()
lib.qs_debug_step_kernargs.argtypes = [C.c_void_p] * 8 + [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
for i in range(P):
    rc = lib.qs_debug_step_kernargs(env._h, p(pool[i]), p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term), buf, 1024,
                                    C.byref(size), C.byref(split), C.byref(tiles))
    assert rc == 0
    ck(x.qsx_set_kernarg(i, buf, int(size.value)), "qsx_set_kernarg")
block = 128 if split.value else 256
grid = tiles.value * 128 if split.value else ((tiles.value + 3) // 4) * 256
print("envs %d tiles %d grid %d block %d  StepArgs %d B" % (n, tiles.value, grid, block, size.value))
names = {0: "none", 1: "agent", 2: "system"}
for acq, rel in ((2, 2), (1, 1), (1, 0), (0, 1), (0, 0), (1, 1), (0, 0)):
    env.reset(); twin.reset()
    torch.cuda.synchronize()
    el = C.c_double()
    ck(x.qsx_run_chain(K, grid, block, acq, rel, C.byref(el)), "qsx_run_chain")
    for k in range(K):
        twin.step(pool[k % P])
    torch.cuda.synchronize()
    a, b = full_state(env), full_state(twin)
    same = np.array_equal(a, b)
    bad = int((a != b).any(axis=1).sum())
    print("acquire %-6s release %-6s : %.2f us per step  (%.2f G env-steps/s)   state after %d steps == HIP-launched twin: %s%s"
          % (names[acq], names[rel], el.value / K, n * K / el.value / 1e3, K, same, "" if same else "  (%d of %d envs differ)" % (bad, n)))
    assert env.step_counter == twin.step_counter
x.qsx_close()
