"""In-kernel timeline of the role-split step kernel in the REAL launch chain (QS_STAMP build, tools/build_stamp.sh):
    QUADSIM_HIP_LIB=quadsim_amd/csrc/libquadsim_hip_stamp.so python tools/stamp_timeline.py [envs] [groups]
Stamps (100 MHz real-time counter, kept in registers until the wave ends) per workgroup and step: chaser wave 0 start,
1 state loads landed, 2 drone step done (before barrier #1), 3 after #1, 4 obs + reward done (before #2), 5 after #2, 6 after
the reset branch, 7 stores drained; target wave 8 + {0 start, 1 loads landed, 2 before #1, 3 after #1, 4 before #2, 5 after #2,
6 stores issued}."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quadsim_amd as qa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = qa._lib.load()
env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=1234, init_range=qa.C3_INIT_RANGE, copy=False)
tiles = (n + 63) // 64
buf = torch.zeros((64, tiles, 16), dtype=torch.int64, device="cuda")
lib.qs_debug_set_stamps.argtypes = [C.c_void_p, C.c_uint64]
assert lib.qs_debug_set_stamps(C.c_void_p(buf.data_ptr()), buf.numel()) == 0
env.reset()
if G > 1:
    env.set_groups(G, threads=True)
pool = env.random_actions(64, step0=0)
h = env._h
p = lambda t: C.c_void_p(t.data_ptr())
args = (p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term), None)
for rep in range(6):                       # the last 64 steps overwrite the earlier ones (slot = step counter mod 64)
    for k in range(64):
        lib.qs_step_groups(h, p(pool[k]), *args)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.float64) * 0.01          # us
s = s[8:56]                                              # steady state
t0 = s[:, :, 0].min(axis=1)                              # first workgroup start of each launch (all groups)
t_end = s[:, :, 7].max(axis=1)
per = np.diff(t0)
print("envs %d groups %d tiles %d (stamped build: the stamps and the vmcnt wait after the state loads cost ~0.7 us per step)" % (n, G, tiles))
print("step period (first start -> next step's first start): median %.2f us  (min %.2f max %.2f)" % (np.median(per), per.min(), per.max()))
print("start spread (first -> last wg start): median %.2f us" % np.median(s[:, :, 0].max(axis=1) - t0))
print("kernel span (first start -> last drained): median %.2f us" % np.median(t_end - t0))
print("gap (last drained -> next step's first start): median %.2f us" % np.median(t0[1:] - t_end[:-1]))
names = ["start->loads landed", "loads->drone step done", "wait at #1", "#1->obs/reward done", "wait at #2", "reset branch", "stores + drain"]
for j, nm in enumerate(names):
    d = s[:, :, j + 1] - s[:, :, j]
    print("  chaser wave %-24s median %.2f  p90 %.2f us" % (nm, np.median(d), np.percentile(d, 90)))
tn = ["start->loads landed", "loads->step+draw done", "wait at #1", "#1->PID done", "wait at #2", "reset + stores issued"]
for j, nm in enumerate(tn):
    d = s[:, :, 8 + j + 1] - s[:, :, 8 + j]
    print("  target wave %-24s median %.2f  p90 %.2f us" % (nm, np.median(d), np.percentile(d, 90)))
d = s[:, :, 7] - s[:, :, 0]; print("  workgroup lifetime (chaser wave)      median %.2f  p90 %.2f us" % (np.median(d), np.percentile(d, 90)))
if G > 1:
    for g in range(G):
        lo, hi = env.group_range(g)
        sl = s[:, lo // 64:(hi + 63) // 64]
        a, b = sl[:, :, 0].min(axis=1), sl[:, :, 7].max(axis=1)
        print("  group %d: span median %.2f us, period %.2f us, gap %.2f us" % (g, np.median(b - a), np.median(np.diff(a)), np.median(a[1:] - b[:-1])))
env.close()
