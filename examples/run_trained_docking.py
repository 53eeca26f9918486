#!/usr/bin/env python3
"""The loop of the reference's run_trained_docking_ppo2.py:36-60 on the MI355X: the shipped PPO2 actor drives
docking-v0 for one episode in N parallel envs.  Five ways to run the same loop:

    python examples/run_trained_docking.py --envs 65536 --mode step      # policy(obs) -> env.step(a), one launch each
    python examples/run_trained_docking.py --envs 65536 --mode stephip   # policy.predict_hip(env, obs) -> env.step(a): the actor as one MFMA kernel
    python examples/run_trained_docking.py --envs 65536 --mode step1     # env.step_policy(policy): actor + step in ONE launch per step
    python examples/run_trained_docking.py --envs 65536 --mode fused     # whole loop in ONE launch, exact-f32 MFMA
    python examples/run_trained_docking.py --envs 65536 --mode fast      # split-bf16 MFMA actor (~1e-5 action error)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import quadsim_amd as qa  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--envs", type=int, default=4096)
p.add_argument("--steps", type=int, default=600)
p.add_argument("--mode", default="fused", choices=["step", "stephip", "step1", "fused", "fast"])
p.add_argument("--jitter", action="store_true", help="rocRAND-randomised initial states (BASELINE config 3 ranges)")
args = p.parse_args()

weights = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "policy_best_model_v0.npz")
policy = qa.MlpPolicy.from_npz(weights)
env = qa.VecDockingEnv("docking-v0", num_envs=args.envs, randomise=1 if args.jitter else 0, seed=0,
                       init_range=qa.C3_INIT_RANGE)
if args.mode in ("fused", "fast"):                    # warm-up launch (LDS weight image, allocator), then start over
    qa.fused_policy_rollout(env, policy, 2, precision="f32" if args.mode == "fused" else "bf16x3")
obs = env.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
if args.mode == "step":
    O, R, D, F, A = qa.rollout_with_policy(env, policy, args.steps, obs0=obs)
elif args.mode == "stephip":                          # the VecEnv protocol as in "step" (infos, terminal observations), actor on the matrix cores
    rows = []
    for _ in range(args.steps):
        a = policy.predict_hip(env, obs)
        obs, r, d, info = env.step(a)
        rows.append((obs, r, d, env.last_flags, a))
    O, R, D, F, A = (torch.stack(x) for x in zip(*rows))
elif args.mode == "step1":                            # outputs consumable after every step, one launch per step
    rows = []
    for _ in range(args.steps):
        o, r, d, a = env.step_policy(policy)
        rows.append((o, r, d, env.last_flags, a))
    O, R, D, F, A = (torch.stack(x) for x in zip(*rows))
else:
    O, R, D, F, A = qa.fused_policy_rollout(env, policy, args.steps, precision="f32" if args.mode == "fused" else "bf16x3")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
docked = (F & 1).bool()
print("%d envs x %d steps in %.1f ms  (%.2f G env-steps/s, mode %s)" % (args.envs, args.steps, dt * 1e3,
                                                                       args.envs * args.steps / dt / 1e9, args.mode))
print("mean return per env %.4f   docked steps per env %.1f   episodes ended %d" % (
    float(R.sum(0).mean()), float(docked.sum(0).float().mean()), int(D.sum())))
env.close()
