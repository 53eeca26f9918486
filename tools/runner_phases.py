"""Phase budget of the role-split Runner kernel (QS_STAMP build, tools/build_stamp.sh):
    QUADSIM_HIP_LIB=quadsim_amd/csrc/libquadsim_hip_stamp.so python tools/runner_phases.py [envs] [T] [f32|bf16x3]
Durations of the phases of the matrix wave and of the env wave of every tile, summed over the T steps of one launch by the
kernel itself (100 MHz counter), printed as the median over tiles in us per step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quadsim_amd as qa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
lib = qa._lib.load()
w = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "policy_best_model_v0.npz")
ac = qa.ActorCriticPolicy.from_npz(w)
env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=0, init_range=qa.C3_INIT_RANGE)
env.reset()
tiles = (n + 63) // 64
buf = torch.zeros((tiles, 2, 8), dtype=torch.int64, device="cuda")
lib.qs_debug_set_stamps.argtypes = [C.c_void_p, C.c_uint64]
assert lib.qs_debug_set_stamps(C.c_void_p(buf.data_ptr()), buf.numel()) == 0
qa.fused_runner_rollout(env, ac, T, precision=prec)
torch.cuda.synchronize()
import time
t0 = time.perf_counter(); qa.fused_runner_rollout(env, ac, T, precision=prec); torch.cuda.synchronize(); wall = time.perf_counter() - t0
raw = buf.cpu().numpy().astype(np.float64)
print("shader clock over the matrix waves' loops: median %.0f MHz" % np.median(raw[:, 0, 6] / raw[:, 0, 7] * 100.0))
s = raw * 0.01 / T          # us per step
print("envs %d  T %d  %s  wall %.2f us/step" % (n, T, prec, wall / T * 1e6))
names_m = ["wait #a", "layer 1", "policy branch + means", "wait #b", "value branch", "value out"]
names_e = ["mb_obs store", "wait #a", "noise draw", "wait #b", "sample + stores", "env step + obs out"]
for r, names in ((0, names_m), (1, names_e)):
    print("matrix wave:" if r == 0 else "env wave:")
    for j, nm in enumerate(names):
        print("  %-24s median %.2f  p90 %.2f us/step" % (nm, np.median(s[:, r, j]), np.percentile(s[:, r, j], 90)))
    print("  %-24s %.2f us/step" % ("sum", np.median(s[:, r, :6].sum(axis=1))))
env.close()
