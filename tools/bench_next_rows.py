#!/usr/bin/env python3
"""Measurement of the SURVEY.md section-8f rows on one MI355X (not the headline bench): hovering-v0 / docking-v1 /
docking-v2 step, GAE + swap_and_flatten, PID expert, policy-in-the-loop.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa

PEAK = 8000.0
N = 65536


def timed(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps      # us per call


out = {}
for env_id, bpe in (("docking-v0", 392), ("docking-v2", 392), ("docking-v1", 392 + 0), ("hovering-v0", 17 * 4 + 16 + 17 * 4 + 52 + 4 + 2)):
    env = qa.VecDockingEnv(env_id, num_envs=N, randomise=1 if env_id in ("docking-v0", "docking-v2") else 0, seed=1,
                           init_range=qa.C3_INIT_RANGE, copy=False)     # the fast path of the Python API: outputs valid until the next step
    env.reset()
    acts = env.random_actions(64)
    if env_id == "hovering-v0":
        acts = acts * 0.5 + 0.5
    k = [0]
    def step():
        env.step_async(acts[k[0] % 64]); k[0] += 1
    us = timed(step, 1000, 100)
    T = 64
    outb = env.rollout(acts)
    us_r = timed(lambda: env.rollout(acts, out=outb), 10, 2) / T
    out[env_id] = {"step_us": us, "step_env_steps_per_s": N / us * 1e6, "bytes_per_env_step": bpe,
                   "hbm_frac": bpe * N / (us * 1e-6) / 1e9 / PEAK, "rollout_T64_env_steps_per_s": N / us_r * 1e6}
    env.close()

env = qa.VecDockingEnv("docking-v0", num_envs=N)
T = 600
g = torch.Generator(device="cuda").manual_seed(0)
rew = torch.randn((T, N), device="cuda", generator=g); val = torch.randn((T, N), device="cuda", generator=g)
dn = torch.rand((T, N), device="cuda", generator=g) < 0.02
lv = torch.randn(N, device="cuda", generator=g); ld = torch.rand(N, device="cuda", generator=g) < 0.1
dn8, ld8 = dn.to(torch.uint8), ld.to(torch.uint8)
us = timed(lambda: qa.compute_gae(env, rew, val, dn8, lv, ld8, 0.99, 0.95), 20, 3)
gae_bytes = T * N * ((4 + 4 + 1) + 8)            # single-pass kernel (N >= 16384): rewards/values/dones read once, advs/returns written
out["gae_T600"] = {"us": us, "elements_per_s": T * N / us * 1e6, "bytes": gae_bytes, "hbm_frac": gae_bytes / (us * 1e-6) / 1e9 / PEAK}
obs = torch.randn((T, N, 12), device="cuda", generator=g)
us = timed(lambda: qa.swap_and_flatten(env, obs), 10, 2)
out["swap_and_flatten_T600_d12"] = {"us": us, "bytes": 2 * obs.numel() * 4, "hbm_frac": 2 * obs.numel() * 4 / (us * 1e-6) / 1e9 / PEAK}
ex = qa.PIDExpert(env)
us = timed(lambda: ex.act(), 500, 20)
out["pid_expert_action"] = {"us": us, "envs_per_s": N / us * 1e6}
pol = qa.MlpPolicy.from_npz(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "policy_best_model_v0.npz"))
obs0 = env.reset()
state = {"obs": obs0}
def loop():
    a = pol.predict(state["obs"])
    state["obs"], _, _, _ = env.step(a)
us = timed(loop, 300, 20)
out["policy_in_the_loop_torch_gemm"] = {"us_per_step": us, "env_steps_per_s": N / us * 1e6,
                                        "mlp_flop_per_env_step": 2 * (12 * 128 + 128 * 128 + 128 * 4)}
# the same loop captured once with torch.cuda.graphs (the step counter lives on the device: replays stay correct)
K = 16
static_obs = state["obs"].clone()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    pol.predict(static_obs)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    o = static_obs
    for _ in range(K):
        o, _, _, _ = env.step(pol.predict(o))
    static_obs.copy_(o)
us = timed(lambda: g.replay(), 30, 3) / K
out["policy_in_the_loop_torch_gemm_hipgraph"] = {"us_per_step": us, "env_steps_per_s": N / us * 1e6, "steps_per_graph": K}
env.close()
for n in (65536, 262144):
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=1, init_range=qa.C3_INIT_RANGE)
    env.reset()
    T = 32
    us = timed(lambda: qa.fused_policy_rollout(env, pol, T, want_actions=False), 5, 2) / T
    flop = 2 * (12 * 128 + 128 * 128 + 128 * 4)
    out["policy_in_the_loop_fused_mfma_N%d" % n] = {"us_per_step": us, "env_steps_per_s": n / us * 1e6,
        "mlp_tflops": n * flop / (us * 1e-6) / 1e12, "mfma_f32_peak_tflops": 157.3,
        "mfma_frac": n * flop / (us * 1e-6) / 1e12 / 157.3}
    us = timed(lambda: qa.fused_policy_rollout(env, pol, T, want_actions=False, precision="bf16x3"), 5, 2) / T
    out["policy_in_the_loop_fused_bf16x3_N%d" % n] = {"us_per_step": us, "env_steps_per_s": n / us * 1e6,
        "note": "split-bf16 (hi+lo) operands on the bf16 matrix rate; ~1e-5 action error (opt-in)"}
    env.close()
print(json.dumps(out, indent=1))
