"""A/B of library builds on the policy-in-the-loop kernels (set QUADSIM_HIP_LIB)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
N, T = 65536, 32
w = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "policy_best_model_v0.npz")
pol = qa.MlpPolicy.from_npz(w); ac = qa.ActorCriticPolicy.from_npz(w)
env = qa.VecDockingEnv("docking-v0", num_envs=N, randomise=1, seed=0, init_range=qa.C3_INIT_RANGE)
env.reset()
def tm(f, reps=6):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps / T * 1e6
res = []
for prec in ("f32", "bf16x3"):
    res.append("policy %s %.2f us/step" % (prec, tm(lambda: qa.fused_policy_rollout(env, pol, T, want_actions=False, precision=prec))))
    res.append("runner %s %.2f us/step" % (prec, tm(lambda: qa.fused_runner_rollout(env, ac, T, precision=prec))))
print(os.environ.get("QUADSIM_HIP_LIB", "default").split("/")[-1], " | ".join(res))
