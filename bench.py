#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused docking env.step() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over the batch: ONE qs_step launch stepping
every env of this GPU once (BASELINE.json config 3: 65 536 docking-v0 envs per
GPU, U(-1,1) random actions already resident in HBM, rocRAND randomised
auto-reset).  Envs shard over GPUs by env id with no data-path collective
(scaling "weak": 65 536 envs per GPU); the RCCL all-gather of roll-out slabs that
BASELINE configs 4/5 mention is timed separately and reported under "allgather",
never mixed into `value`.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (HBM, algorithmic
bytes / launch), "cpu_baseline" (the C oracle timed on the host cores),
"rollout_fused" (qs_rollout: T steps per launch, state in registers).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec peak
BYTES_PER_ENV_STEP = 392       # read 176 B + write 216 B per env-step (step-API mode)
BYTES_PER_ENV_STEP_DR = 408    # + mass, Ixx, Iyy, Izz read (per-env params)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--envs-per-gpu", type=int, default=65536)
    p.add_argument("--env", default="docking-v0", choices=["docking-v0", "docking-v2"])
    p.add_argument("--integrator", default="frozen", choices=["frozen", "rk4"])
    p.add_argument("--randomise", type=int, default=1, help="0 nominal resets, 1 rocRAND init state, 2 + mass/inertia")
    p.add_argument("--action-pool", type=int, default=512, help="distinct pre-generated [N,4] action batches cycled")
    p.add_argument("--rollout-T", type=int, default=64, help="steps per launch of the fused roll-out leg")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the rollout_fused / rk4 / allgather legs")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--repeats", type=int, default=5, help="runs of (W untimed + K timed) steps; the median is reported")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl (= RCCL over xGMI, the real thing) or gloo (rehearsal of the multi-rank control flow, "
                        "e.g. several ranks sharing one GPU; skips the all-gather leg)")
    return p.parse_args()


def cpu_baseline(kind, seconds, seed=1234):
    """The C oracle (oracle/quadsim_oracle.c, f64 build == the reference's float64 arithmetic restated)
    timed on the host cores, one thread per core (ctypes releases the GIL), same workload shape:
    U(-1,1) actions, auto-reset with randomised init.  Bounded sample."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from oracle.pyoracle import Oracle, PAR_NOMINAL
    orc = Oracle("f64")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    n, T = 1024, 50
    rr = (0.5, 0.1, 0.2, 0.1, 1.0, 1.0, 1.0, 1.0)
    rs = np.random.RandomState(seed)
    acts = rs.uniform(-1, 1, (T, n, 4))

    def worker(w):
        rec = orc.env_init(n)
        par = np.tile(np.array(PAR_NOMINAL, np.float64), (n, 1))
        orc.vec_reset(rec, par, randomise=1, seed=seed, gid0=w * n, rr=rr)
        done_steps, k = 0, 0
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            orc.vec_rollout(rec, par, acts, kind=kind, randomise=1, seed=seed, step_idx0=k, gid0=w * n, rr=rr)
            k += T
            done_steps += n * T
        return done_steps

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "C oracle (f64 restatement of the reference path), %d threads x 1024 envs, U(-1,1) actions, "
                      "randomised auto-reset, %.1f s wall" % (cores, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus=%d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()          # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or bool(os.environ.get("QS_BENCH_FORCE_DIST"))   # the env hook rehearses the RCCL path on one GPU
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import ctypes as C
    from quadsim_amd import C3_INIT_RANGE, VecDockingEnv, _lib, shard_range

    n = args.envs_per_gpu
    total_envs = n * world
    lo, hi = shard_range(total_envs, rank, world)
    assert hi - lo == n
    kind = 0 if args.env == "docking-v0" else 1

    def make_env(integrator):
        return VecDockingEnv(args.env, num_envs=n, device=local_rank, integrator=integrator, randomise=args.randomise,
                             seed=1234, env_id_offset=lo, init_range=C3_INIT_RANGE, mass_scale=(0.8, 1.2),
                             inertia_scale=(0.8, 1.2))

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def time_steps(env, K, W, pool):
        """W untimed + K timed qs_step launches; returns (wall seconds max over ranks, HIP-event ms)"""
        lib, h = env._lib, env._h
        P = pool.shape[0]
        aptr = [C.c_void_p(pool[i].data_ptr()) for i in range(P)]
        obs, rew, done, flags, term = (env._ptr(env._obs), env._ptr(env._rew), env._ptr(env._done),
                                       env._ptr(env._flags), env._ptr(env._term))
        step = lib.qs_step
        for k in range(W):
            step(h, aptr[k % P], obs, rew, done, flags, term)
        barrier()
        env.timer_start()
        t0 = time.perf_counter()
        for k in range(K):
            step(h, aptr[k % P], obs, rew, done, flags, term)
        ev_ms = env.timer_stop()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if distributed:
            dist.barrier()
        return max_over_ranks(wall), ev_ms

    K, W = args.steps, args.warmup
    env = make_env(args.integrator)
    env.reset()
    P = max(1, min(args.action_pool, K))
    pool = env.random_actions(P, step0=0)               # [P,N,4] U(-1,1), resident in HBM before timing
    # median of 5 runs.  Every run is W untimed + EXACTLY K timed steps between barrier + sync;
    # the reported run is the median one (its wall clock AND its HIP-event time), all five are listed in "runs".
    runs = [time_steps(env, K, W, pool) for _ in range(max(1, args.repeats))]
    order = sorted(range(len(runs)), key=lambda i: runs[i][0])
    wall, ev_ms = runs[order[len(order) // 2]]
    value = total_envs * K / wall
    bpe = BYTES_PER_ENV_STEP_DR if args.randomise >= 2 else BYTES_PER_ENV_STEP
    launch_us = ev_ms * 1e3 / K
    achieved = bpe * n / (launch_us * 1e-6) / 1e9       # GB/s per GPU, algorithmic bytes / avg launch period

    out = {
        "metric": "env-steps/sec at N parallel envs (1/2/4/8 GPU); per-step state L2 err vs NumPy ref",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "runs": {"policy": "median of %d runs of W untimed + K timed steps" % len(runs),
                 "env_steps_per_s": [total_envs * K / r[0] for r in runs]},
        "config": {"workload": "BASELINE config 3: %d parallel %s envs per GPU, U(-1,1) random actions, SoA tiles + "
                               "rocRAND randomised auto-reset" % (n, args.env),
                   "envs_per_gpu": n, "total_envs": total_envs, "env": args.env, "integrator": args.integrator,
                   "dt": 0.02, "randomise": args.randomise, "mode": "step-API (one qs_step launch per step)",
                   "parallelism": "env-sharded x%d, no data-path collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": ("k_env_split<%s> (T=1): two waves per tile" if n <= 131072 else "k_env<%s> (T=1)") % args.integrator,
                     "bytes_per_env_step": bpe,
                     "bytes_per_launch": bpe * n, "launch_period_us": launch_us,
                     "read_frac": (176 if bpe == 392 else 192) * n / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "basis": "HIP events on the launching stream around the K timed launches / K "
                              "(launch period incl. inter-kernel gaps); working set %.1f MB is Infinity-Cache resident"
                              % (n * 160 / 1e6)},
    }

    # HBM-side bytes per launch from the PMC passes (collected separately: rocprofv3 --pmc cannot run inside
    # this process); only quoted when the profile was taken on this very configuration
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))
        if pmc["envs"] == n and args.env == "docking-v0" and args.integrator == "frozen" and args.randomise == 1:
            out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = pmc["source"]
    except (OSError, ValueError, KeyError):
        pass

    if not args.no_extras:
        # fused roll-out leg: T steps per launch, state in registers (different algorithmic bytes: see DESIGN.md)
        T = args.rollout_T
        acts = pool[:T] if P >= T else env.random_actions(T)
        reps = max(1, min(20, K // T))
        env.rollout(acts)
        barrier()
        env.timer_start()
        t0 = time.perf_counter()
        for _ in range(reps):
            o_, r_, d_, f_ = env.rollout(acts, want_flags=False)
        ms = env.timer_stop()
        torch.cuda.synchronize()
        w2 = max_over_ranks(time.perf_counter() - t0)
        b_roll = 16 + 48 + 4 + 1 + 320.0 / T
        out["rollout_fused"] = {"value": total_envs * T * reps / w2, "unit": "env-steps/s", "T": T, "launches": reps,
                                "bytes_per_env_step": b_roll,
                                "hbm_frac": b_roll * n * T * reps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "qs_rollout: identical results to T qs_step calls; VALU-bound, not HBM-bound"}
        del o_, r_, d_, f_
        # rk4 integrator (the physically intended mode; same kernel, 4 df evaluations per drone)
        other = "rk4" if args.integrator == "frozen" else "frozen"
        env_o = make_env(other)
        env_o.reset()
        Ko = max(100, K // 4)
        w3, ms3 = time_steps(env_o, Ko, min(W, 50), pool)
        out["other_integrator"] = {"integrator": other, "value": total_envs * Ko / w3, "unit": "env-steps/s",
                                   "launch_period_us": ms3 * 1e3 / Ko}
        env_o.close()
        # the same step kernel where it is not latency-bound: 1 048 576 envs per GPU (168 MB of state, 16 waves per
        # SIMD): shows the kernel's bandwidth ceiling beside the 65 536-env headline
        if n < (1 << 20):
            nb = 1 << 20
            env_b = VecDockingEnv(args.env, num_envs=nb, device=local_rank, integrator=args.integrator,
                                  randomise=args.randomise, seed=1234, env_id_offset=rank * nb, init_range=C3_INIT_RANGE,
                                  mass_scale=(0.8, 1.2), inertia_scale=(0.8, 1.2))
            env_b.reset()
            pool_b = env_b.random_actions(8, step0=0)
            wb, msb = time_steps(env_b, 200, 20, pool_b)
            out["step_api_1M_envs"] = {"envs_per_gpu": nb, "value": nb * world * 200 / wb, "unit": "env-steps/s",
                                       "launch_period_us": msb * 1e3 / 200,
                                       "hbm_frac": bpe * nb / (msb * 1e-3 / 200) / 1e9 / HBM_PEAK_GBS}
            env_b.close()
            del pool_b
        # policy in the loop: a = clip(MLP(obs)); env.step(a), T steps per launch, the shipped PPO2
        # actor (weights fixture) on the matrix cores.  Needs nominal / rocRAND-initialised resets and no per-env params.
        wpath = os.path.join(ROOT, "tests", "golden", "policy_best_model_v0.npz")
        if args.randomise <= 1 and os.path.exists(wpath):
            from quadsim_amd import MlpPolicy, fused_policy_rollout
            pol = MlpPolicy.from_npz(wpath, device="cuda:%d" % local_rank)
            Tp = 32
            flop = 2 * (12 * 128 + 128 * 128 + 128 * 4)
            for prec, key in (("f32", "policy_rollout_f32_mfma"), ("bf16x3", "policy_rollout_bf16x3_mfma")):
                fused_policy_rollout(env, pol, Tp, want_actions=False, precision=prec)
                barrier()
                t0 = time.perf_counter()
                for _ in range(4):
                    fused_policy_rollout(env, pol, Tp, want_actions=False, precision=prec)
                torch.cuda.synchronize()
                w5 = max_over_ranks(time.perf_counter() - t0)
                out[key] = {"value": total_envs * Tp * 4 / w5, "unit": "env-steps/s", "T": Tp,
                            "mlp_tflops": total_envs * Tp * 4 * flop / w5 / 1e12}
            out["policy_rollout_f32_mfma"]["roofline"] = {
                "bound": "mfma", "peak": 157.3 * world, "unit": "TFLOP/s",
                "achieved": out["policy_rollout_f32_mfma"]["mlp_tflops"],
                "frac": out["policy_rollout_f32_mfma"]["mlp_tflops"] / (157.3 * world),
                "note": "exact-f32 MFMA (v_mfma_f32_16x16x4_f32); 36 864 MLP flop per env-step"}
            out["policy_rollout_bf16x3_mfma"]["note"] = "split-bf16 operands, 3 MFMAs per product, ~1e-5 action error (opt-in)"
            # PPO2 data collection (Runner._run, rl_baselines/ppo2/ppo2.py:472-527): actor + critic + Gaussian sampling +
            # neglogp + env.step for n_steps in one launch, then the GAE kernel and the env-major flatten of 7 arrays
            from quadsim_amd import ActorCriticPolicy, Runner
            ac = ActorCriticPolicy.from_npz(wpath, device="cuda:%d" % local_rank)
            flop_ac = 2 * (12 * 128 + 2 * 128 * 128 + 128 * 4 + 128)
            for prec, key in (("f32", "ppo2_runner_f32_mfma"), ("bf16x3", "ppo2_runner_bf16x3_mfma")):
                runner = Runner(env=env, model=ac, n_steps=Tp, gamma=0.99, lam=0.95, track_episodes=False, precision=prec)
                runner.run()
                barrier()
                t0 = time.perf_counter()
                for _ in range(4):
                    runner.run()
                torch.cuda.synchronize()
                w6 = max_over_ranks(time.perf_counter() - t0)
                out[key] = {"value": total_envs * Tp * 4 / w6, "unit": "env-steps/s", "T": Tp,
                            "mlp_tflops": total_envs * (Tp + 1) * 4 * flop_ac / w6 / 1e12}
            out["ppo2_runner_f32_mfma"]["what"] = (
                "Runner.run(): qs_runner_rollout (policy + value nets on exact-f32 MFMA, rocRAND Gaussian sampling, "
                "neglogp, fused env.step) + qs_gae + swap_and_flatten of obs/returns/dones/actions/values/neglogp/rewards")
            out["ppo2_runner_bf16x3_mfma"]["note"] = "the same with split-bf16 operands (qs_runner_rollout_fast), ~1e-5 error on means / values (opt-in)"
        if distributed and args.backend == "nccl":
            # BASELINE configs 4/5: RCCL all-gather of the roll-out slabs (obs, reward, done) once per T-step roll-out
            from quadsim_amd.distributed import gather_slab
            slab = env.rollout_slab(acts)
            gathered = gather_slab(slab)
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                env.rollout_slab(acts, out=slab)
                gather_slab(slab, out=gathered)
            torch.cuda.synchronize()
            w4 = max_over_ranks(time.perf_counter() - t0)
            out["allgather"] = {"value": total_envs * T * reps / w4, "unit": "env-steps/s",
                                "what": "qs_rollout_slab(T=%d) + ONE RCCL all_gather of the packed (obs, reward, done) slab "
                                        "(%.1f MB per rank per roll-out)" % (T, T * n * 56 / 1e6)}

    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(kind, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    env.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
