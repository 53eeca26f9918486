#!/usr/bin/env python3
"""Round-3 probe of the stream-ordered private queues (qs_set_queue_ordering).

Per-step loops `obs -> policy -> env.step` in the three launch paths (HIP stream | private queue, stream-ordered |
private queue, host-ordered), with three policies (none: pre-staged actions; one elementwise torch op; the shipped
MlpPolicy), checked bit for bit against the HIP-stream loop, and the pre-staged roll-out (qs_rollout_stepwise).
Prints one JSON object.  Run on the GPU box:  python tools/interop_probe.py [--envs 65536] [--steps 300]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--queues", type=int, default=1)
    ap.add_argument("--sections", default="policy,raw,rollout")
    ap.add_argument("--T", type=int, default=256)
    args = ap.parse_args()
    sections = args.sections.split(",")
    import torch
    import quadsim_amd as qa
    n, K = args.envs, args.steps
    wpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "policy_best_model_v0.npz")
    pol = qa.MlpPolicy.from_npz(wpath, device="cuda:0")

    def make():
        e = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=3, init_range=qa.C3_INIT_RANGE, copy=False)
        e.reset()
        return e

    policies = {
        "prestaged": None,
        "one_op": lambda o: torch.tanh(o[:, 3:7] * 0.7),
        "mlp": lambda o: pol.predict(o),
    }
    out = {"envs": n, "steps": K, "queues": args.queues, "QS_CHAIN_OUT_WT": os.environ.get("QS_CHAIN_OUT_WT", "0")}
    ref = {}
    for mode in (("hip", "private_stream", "private_host") if "policy" in sections else ()):
        for pname, pfn in policies.items():
            env = make()
            if mode != "hip":
                env.set_queue_mode(True, args.queues, ordering="stream" if mode == "private_stream" else "host")
            pool = env.random_actions(16, step0=0)
            obs = env.reset()
            acc = torch.zeros((), dtype=torch.float64, device="cuda")
            for k in range(30):                                   # warm-up: allocator, hipBLASLt heuristics
                a = pool[k % 16] if pfn is None else pfn(obs)
                obs, rew, done, _ = env.step(a)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                a = pool[k % 16] if pfn is None else pfn(obs)
                obs, rew, done, _ = env.step(a)
                if k % 50 == 49:
                    acc = acc + obs.double().sum() + rew.double().sum() + done.double().sum()
            torch.cuda.synchronize()
            env.sync()
            dt = time.perf_counter() - t0
            st = env.get_state(as_numpy=False)
            sig = (float(acc.item()), float(sum(v.double().sum().item() for v in st.values())), env.step_counter)
            key = pname
            if mode == "hip":
                ref[key] = (sig, obs.clone(), {k_: v.clone() for k_, v in st.items()})
                same = True
            else:
                same = bool(sig == ref[key][0] and torch.equal(obs, ref[key][1]) and all(torch.equal(st[k_], ref[key][2][k_]) for k_ in st))
            out["%s/%s" % (mode, pname)] = {"us_per_step": dt / K * 1e6, "identical_to_hip": same}
            env.close()
    # raw qs_step loop from ctypes (what bench.py times): the hand-shake's cost per step without Python's VecEnv plumbing
    import ctypes as C
    for mode in (("hip", "private_host", "private_stream") if "raw" in sections else ()):
        for q in ((0,) if mode == "hip" else (1, 2)):
            env = make()
            if mode != "hip":
                env.set_queue_mode(True, q, ordering="stream" if mode == "private_stream" else "host")
            pool = env.random_actions(16, step0=0)
            lib, h = env._lib, env._h
            pp = lambda t: C.c_void_p(t.data_ptr())            # noqa: E731
            io = (pp(env._obs), pp(env._rew), pp(env._done), pp(env._flags), pp(env._term))
            ap_ = [pp(pool[i]) for i in range(16)]
            seq = [ap_[k % 16] for k in range(2000)]
            for a in seq[:200]:
                lib.qs_step(h, a, *io)
            torch.cuda.synchronize(); env.sync()
            t0 = time.perf_counter()
            for a in seq:
                lib.qs_step(h, a, *io)
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize(); env.sync()
            dt = time.perf_counter() - t0
            out["raw/%s/q%d" % (mode, q)] = {"us_per_step": dt / 2000 * 1e6, "host_issue_us_per_step": t_issue / 2000 * 1e6}
            env.close()
    # pre-staged roll-out: T single-step launches per call
    for mode in (("hip", "private_stream", "private_host") if "rollout" in sections else ()):
        env = make()
        if mode != "hip":
            env.set_queue_mode(True, args.queues, ordering="stream" if mode == "private_stream" else "host")
        T = args.T
        acts = env.random_actions(T, step0=0)
        bufs = env.rollout(acts, stepwise=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            env.rollout(acts, stepwise=True, out=bufs)
        torch.cuda.synchronize()
        env.sync()
        dt = time.perf_counter() - t0
        out["%s/rollout_stepwise_T%d" % (mode, T)] = {"us_per_step": dt / (4 * T) * 1e6, "obs_sum": float(bufs[0].double().sum().item())}
        env.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
