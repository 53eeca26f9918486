"""soak: 1 M envs x 3 000 random-action steps (fused roll-outs, rocRAND resets + per-episode mass / inertia), every
observation / reward / state finite and inside the env's physical bounds; docking-v0 and -v2, frozen and rk4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
for env_id, integ, n_envs in (("docking-v0", "frozen", 1 << 20), ("docking-v2", "frozen", 1 << 20), ("docking-v0", "rk4", 1 << 20),
                              ("docking-v0", "frozen", 1 << 16), ("docking-v2", "rk4", 1 << 17)):      # the last two run the role-split kernel
    env = qa.VecDockingEnv(env_id, num_envs=n_envs, integrator=integ, randomise=2, seed=123, init_range=qa.C3_INIT_RANGE,
                           mass_scale=(0.8, 1.2), inertia_scale=(0.8, 1.2))
    env.reset()
    T, eps, bad = 100, 0, 0
    rmax = 3.0 if env_id == "docking-v0" else 10.0
    t0 = time.perf_counter()
    for it in range(30):
        O, R, D, F = env.rollout(T=T)
        bad += int((~torch.isfinite(O)).sum()) + int((~torch.isfinite(R)).sum())
        # an observation returned after a step is either a fresh reset obs or inside the over-limit radius... or terminal
        npos = O[..., 0:3].norm(dim=-1)
        assert float(npos[D == 0].max()) < rmax + 1e-3
        assert float(O[..., 6].abs().max()) <= 1.5708 + 1e-4 and float(O[..., 8].abs().max()) <= 3.1416 + 1e-4
        eps += int(D.sum())
    st = env.get_state()
    import numpy as np
    fin = all(np.isfinite(st[k]).all() for k in ("chaser", "target", "u_prev", "qdes", "last_shaping", "t"))
    qn = np.linalg.norm(st["chaser"][:, 6:10], axis=1)
    print("%s %s N=%d: %d env-steps in %.1f s, %d episodes, non-finite outputs %d, state finite %s, |q| in [%.4f, %.4f], t max %d"
          % (env_id, integ, n_envs, n_envs * T * 30, time.perf_counter() - t0, eps, bad, fin, qn.min(), qn.max(), st["t"].max()))
    env.close()
