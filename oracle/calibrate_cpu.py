#!/usr/bin/env python3
"""CPU calibration: how much faster is the C oracle (the "port" that
bench.py times on the GPU box as `cpu_baseline`) than the TRUE reference, on the same core?

Runs ONLY in the build container: it imports the real reference from /root/reference (through oracle/gen_goldens.py's
loader), steps its DockingEnv / MovingDockingEnv with U(-1,1) actions on ONE core, then times the C oracle on ONE core on
the same workload shape as bench.py's cpu_baseline leg.  Writes oracle/cpu_calibration.json (tracked; numbers only), from
which bench.py derives `cpu_baseline.reference_equivalent` = port rate / ratio: what the reference itself would reach
on the GPU box's host cores.  The reference never leaves this container.

    python oracle/calibrate_cpu.py [seconds-per-leg]
"""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)


def time_reference(env_cls, seconds, seed=0):
    env = env_cls()
    rs = np.random.RandomState(seed)
    env.reset()
    steps, episodes = 0, 0
    # warm-up (imports, first RK45 construction)
    for _ in range(50):
        _, _, d, _ = env.step(rs.uniform(-1, 1, 4).astype(np.float32))
        if d:
            env.reset()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(100):
            _, _, d, _ = env.step(rs.uniform(-1, 1, 4).astype(np.float32))
            steps += 1
            if d:
                env.reset()
                episodes += 1
    return steps / (time.perf_counter() - t0), episodes


def time_port(kind, seconds, seed=1234):
    from oracle.pyoracle import Oracle, PAR_NOMINAL
    orc = Oracle("f64")
    n, T = 1024, 50
    rr = (0.5, 0.1, 0.2, 0.1, 1.0, 1.0, 1.0, 1.0)
    acts = np.random.RandomState(seed).uniform(-1, 1, (T, n, 4))
    rec = orc.env_init(n)
    par = np.tile(np.array(PAR_NOMINAL, np.float64), (n, 1))
    orc.vec_reset(rec, par, randomise=1, seed=seed, gid0=0, rr=rr)
    orc.vec_rollout(rec, par, acts, kind=kind, randomise=1, seed=seed, step_idx0=0, gid0=0, rr=rr)
    k, done_steps = T, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.vec_rollout(rec, par, acts, kind=kind, randomise=1, seed=seed, step_idx0=k, gid0=0, rr=rr)
        k += T
        done_steps += n * T
    return done_steps / (time.perf_counter() - t0)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})      # one core for both legs
    except (AttributeError, OSError):
        pass
    import gen_goldens as gg                       # loads the reference (stub gym / evdev namespaces, env files by path)
    out = {"what": "one core, same container: C oracle (f64 port) vs the imported NumPy/SciPy reference, U(-1,1) actions, "
                   "reset on done; ratio = port / reference", "seconds_per_leg": seconds,
           "host": {"machine": platform.machine(), "python": platform.python_version(), "numpy": np.__version__}}
    try:
        import scipy
        out["host"]["scipy"] = scipy.__version__
        with open("/proc/cpuinfo") as f:
            out["host"]["cpu"] = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "?")
    except (ImportError, OSError):
        pass
    for name, cls, kind in (("docking-v0", gg.env_v0.DockingEnv, 0), ("docking-v2", gg.env_v2.MovingDockingEnv, 1)):
        ref, eps = time_reference(cls, seconds)
        port = time_port(kind, seconds)
        out[name] = {"reference_env_steps_per_s": ref, "reference_episodes": eps, "port_env_steps_per_s": port, "ratio": port / ref}
        print("%s: reference %.1f env-steps/s (%d episodes), C oracle %.3g env-steps/s, ratio %.0f" % (name, ref, eps, port, port / ref))
    path = os.path.join(HERE, "cpu_calibration.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
