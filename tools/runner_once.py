"""a few qs_runner_rollout launches (T = 32, 65 536 envs, f32 and bf16x3) for rocprofv3 --pmc passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
ac = qa.ActorCriticPolicy.from_npz(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "policy_best_model_v0.npz"))
env = qa.VecDockingEnv("docking-v0", num_envs=65536, randomise=1, seed=0, init_range=qa.C3_INIT_RANGE)
env.reset()
for prec in ("f32", "bf16x3"):
    for _ in range(4):
        qa.fused_runner_rollout(env, ac, 32, precision=prec)
torch.cuda.synchronize()
env.close()
