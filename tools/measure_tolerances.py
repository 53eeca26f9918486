#!/usr/bin/env python3
"""Observed errors behind the three tolerances VERDICT round 2 called loose (tests/test_gpu_parity.py controller moments and
closed-loop shim, __graft_entry__.smoke reward): prints the measured maxima so that each tolerance can be set to ~2x of it."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quadsim_amd as qa


def gold(name):
    with np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")) as z:
        return {k: z[k] for k in z.files}


g = gold("g3_controller")
u, sd = qa.ctrl_batch(0, g["state_des"], g["state_now"], mass=float(g["mass"]))
e = np.abs(u - g["u_pid"])
print("g3 PID   : max |du| thrust %.3g, moments %.3g (|u| up to %.3g); rel-to-tol(1e-5 rtol) excess %.3g"
      % (e[:, 0].max(), e[:, 1:].max(), np.abs(g["u_pid"]).max(), (e - 1e-5 * np.abs(g["u_pid"])).max()))
u, sd = qa.ctrl_batch(1, g["state_des"], g["state_now"], g["state_last"], mass=float(g["mass"]))
e = np.abs(u - g["u_vel"])
print("g3 vel   : max |du| thrust %.3g, moments %.3g; excess over rtol 1e-5: %.3g" % (e[:, 0].max(), e[:, 1:].max(), (e - 1e-5 * np.abs(g["u_vel"])).max()))
for name, env_id in (("g4_traj_v0", "docking-v0"), ("g4_traj_v2", "docking-v2")):
    g = gold(name)
    env = qa.make(env_id)
    obs = env.reset()
    eo, er = 0.0, 0.0
    for t in range(400):
        obs, rew, done, info = env.step(g["actions"][t])
        eo = max(eo, float(np.max(np.abs(obs - g["obs"][t]) - 1e-3 * 0 )))
        er = max(er, abs(rew - g["reward"][t]))
        if done:
            obs = env.reset()
    env.close()
    print("closed loop %s: max |obs - ref| over 400 free-running steps %.3g, max |reward - ref| %.3g" % (env_id, eo, er))
# smoke reward
import torch
from oracle.pyoracle import PAR_NOMINAL, Oracle, rec_pack
n, seed = 256, 7
rr = tuple(qa.C3_INIT_RANGE) + (1.0, 1.0, 1.0, 1.0)
env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=seed, init_range=qa.C3_INIT_RANGE)
orc = Oracle("f64")
env.reset()
t0 = np.zeros(n, np.float32); t0[:64] = 597.0
env.set_state(t=t0)
wr, wo = 0.0, 0.0
for k in range(8):
    st = env.get_state()
    rec = rec_pack(st["chaser"], st["target"], st["u_prev"][:, :4], st["u_prev"][:, 4:], st["qdes"], st["last_shaping"], st["t"], dtype=np.float64)
    par = np.tile(np.array(PAR_NOMINAL, np.float64), (n, 1))
    a = env.random_actions(1)[0]
    kk = env.step_counter
    obs, rew, done, infos = env.step(a)
    o_ref, r_ref, d_ref, f_ref, _ = orc.vec_step(rec, par, a.cpu().numpy(), kind=0, randomise=1, seed=seed, step_idx=kk, rr=rr)
    wr = max(wr, float(np.max(np.abs(rew.cpu().numpy() - r_ref))))
    wo = max(wo, float(np.max(np.abs(obs.cpu().numpy() - o_ref))))
print("smoke: max |reward - oracle| %.3g, max |obs - oracle| %.3g" % (wr, wo))
