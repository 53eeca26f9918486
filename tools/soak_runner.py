"""soak of the PPO2 Runner kernels: 40 x Runner.run() (65 536 envs x 600 steps, f32 and split-bf16, nominal and
domain-randomised resets); every returned tensor finite, episode accounting consistent"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
w = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "policy_best_model_v0.npz")
model = qa.ActorCriticPolicy.from_npz(w)
for prec, rnd in (("f32", 0), ("bf16x3", 1), ("f32", 2)):
    env = qa.VecDockingEnv("docking-v0", num_envs=65536, randomise=rnd, seed=3, init_range=qa.C3_INIT_RANGE,
                           mass_scale=(0.9, 1.1), inertia_scale=(0.9, 1.1))
    r = qa.Runner(env=env, model=model, n_steps=600, gamma=0.99, lam=0.95, collect_ep_infos=False, precision=prec)
    bad = 0; eps = 0; t0 = time.perf_counter()
    for it in range(40 if rnd == 0 else 10):
        out = r.run()
        print("  ", prec, rnd, "run", it, "issued", flush=True)
        for x in (out[0], out[1], out[3], out[4], out[5], out[8]):
            bad += int((~torch.isfinite(x)).sum())
        eps += r.last_ep_returns.numel()
    torch.cuda.synchronize()
    print("%s randomise=%d: %d samples, %d episodes, non-finite %d, mean return %.3f, %.1f s" % (
        prec, rnd, r.num_timesteps, eps, bad, float(r.last_ep_returns.mean()), time.perf_counter() - t0))
    env.close()
