"""Per-step cost of a policy-in-the-loop step with per-step consumable outputs, three ways (65 536 docking-v0 envs, 300 steps):
  (a) VecDockingEnv.step(MlpPolicy.predict(obs))     three torch GEMMs + the step kernel (bench.py `policy_between_steps`)
  (b) qs_policy_rollout(T = 1) per step              ONE launch per step: the actor on exact-f32 MFMA fused with the env step
  (c) qs_policy_rollout_fast(T = 1) per step         the same with split-bf16 operands
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
from quadsim_amd import _lib
from quadsim_amd.policy import MlpPolicy, pack_fast_weights

n, K = 65536, 300
pol = MlpPolicy.from_npz(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "policy_best_model_v0.npz"))
env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=1, init_range=qa.C3_INIT_RANGE, copy=False)
obs = env.reset()
for _ in range(20):
    obs, r, d, _i = env.step(pol.predict(obs))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    obs, r, d, _i = env.step(pol.predict(obs))
torch.cuda.synchronize()
print("(a) torch GEMMs + qs_step      : %.2f us per step" % ((time.perf_counter() - t0) / K * 1e6))
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
wt = [pol.w0.t().contiguous(), pol.b0.contiguous(), pol.w1.t().contiguous(), pol.b1.contiguous(), pol.w2.t().contiguous(), pol.b2.contiguous()]
O = torch.empty((1, n, 12), device="cuda"); R = torch.empty((1, n), device="cuda")
D = torch.empty((1, n), dtype=torch.uint8, device="cuda"); F = torch.empty((1, n), dtype=torch.uint8, device="cuda")
A = torch.empty((1, n, 4), device="cuda")
lib, h = env._lib, env._h
env._use_current_stream()
for want in (True, False):
    a = p(A) if want else None
    for _ in range(20):
        lib.qs_policy_rollout(h, 1, *[p(w) for w in wt], p(O), p(R), p(D), p(F), a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        _lib.check(lib.qs_policy_rollout(h, 1, *[p(w) for w in wt], p(O), p(R), p(D), p(F), a))
    torch.cuda.synchronize()
    print("(b) qs_policy_rollout T=1 %s: %.2f us per step" % ("(actions kept)" if want else "(no actions)  ", (time.perf_counter() - t0) / K * 1e6))
blob = torch.as_tensor(pack_fast_weights(pol).copy()).to("cuda")
for _ in range(20):
    lib.qs_policy_rollout_fast(h, 1, p(blob), p(O), p(R), p(D), p(F), p(A))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    _lib.check(lib.qs_policy_rollout_fast(h, 1, p(blob), p(O), p(R), p(D), p(F), p(A)))
torch.cuda.synchronize()
print("(c) qs_policy_rollout_fast T=1 : %.2f us per step" % ((time.perf_counter() - t0) / K * 1e6))
for T in (8, 64):
    O2 = torch.empty((T, n, 12), device="cuda"); R2 = torch.empty((T, n), device="cuda")
    D2 = torch.empty((T, n), dtype=torch.uint8, device="cuda"); F2 = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    lib.qs_policy_rollout(h, T, *[p(w) for w in wt], p(O2), p(R2), p(D2), p(F2), None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        lib.qs_policy_rollout(h, T, *[p(w) for w in wt], p(O2), p(R2), p(D2), p(F2), None)
    torch.cuda.synchronize()
    print("    qs_policy_rollout T=%d     : %.2f us per step" % (T, (time.perf_counter() - t0) / (10 * T) * 1e6))
env.close()
