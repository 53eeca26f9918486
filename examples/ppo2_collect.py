#!/usr/bin/env python3
"""PPO2 data collection on the GPU: what `Runner.run()` of rl_baselines/ppo2/ppo2.py does for run_docking_ppo2.py's
10 subprocess envs x 600 steps, here for 65 536 envs x 600 steps in a handful of kernel launches (policy + value
networks, Gaussian sampling, env.step, GAE, flatten).  The returned tensors are the minibatch source of PPO2's update."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import quadsim_amd as qa  # noqa: E402

N, T = 65536, 600                                     # n_steps = 600 as in the shipped model's hyper-parameters
weights = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "policy_best_model_v0.npz")
model = qa.ActorCriticPolicy.from_npz(weights)        # trained_model/best_model_v0.zip: actor, critic, logstd
env = qa.VecDockingEnv("docking-v0", num_envs=N)     # nominal resets, as in the reference's training runs
runner = qa.Runner(env=env, model=model, n_steps=T, gamma=0.99, lam=0.95, collect_ep_infos=False)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    obs, returns, masks, actions, values, neglogpacs, states, ep_infos, true_reward = runner.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n_ep = runner.last_ep_returns.numel()
    print("run %d: %d samples in %.3f s = %.2f G env-steps/s | %d episodes ended, mean return %.3f, mean length %.1f, "
          "mean value %.3f" % (it, obs.shape[0], dt, obs.shape[0] / dt / 1e9, n_ep,
                               float(runner.last_ep_returns.mean()) if n_ep else float("nan"),
                               float(runner.last_ep_lengths.float().mean()) if n_ep else float("nan"), float(values.mean())))
env.close()
