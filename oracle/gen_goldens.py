#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz from the REAL reference.

Runs ONLY in the build container (it imports /root/reference at run time; the
reference never travels to the GPU box).  The fixtures are data: inputs and the
reference's outputs.  No reference source is copied anywhere.

Loading recipe: the reference's packages pull optional
third-party modules that are absent here (gym, evdev, zmq), so tiny stub
namespaces are pre-seeded in sys.modules and the two env files are loaded by
file path.  Nothing of the reference's arithmetic is stubbed.

Fixture sets:
  g1_drone_step     Drone.step single steps, incl. attitude-limiter hits and thrust clamps
  g2_transforms     quat2euler / euler2quat / quat2rot / rot2euler known answers
  g3_controller     controller.PID / vel_controller single calls (state_des before/after)
  g4_traj_v0/_v2    env.step trajectories with the full internal state every step
  g5_policy_episode v0 episode driven by trained_model/best_model_v0.zip (reaches the docked state)
  g6_sim_pid        run_sim_PID.py:8-54 hover loop, 2000 steps (BASELINE config 1)
  g7_domain_rand    v0/v2 trajectories with patched mass / inertia
  g8_traj_v1        docking-v1 (construction-time chaser jitter) trajectories, three constructions
  g12_dock_port     Drone.get_dock_port_state for random states / ports
  g11_expert_episode PID expert (vel_controller on the chaser + inverse action map) on docking-v0
  g10_gae           GAE(lambda) + swap_and_flatten of the in-tree PPO2 Runner (reference lines executed)
  g9_hovering       hovering-v0 trajectories (raw-state obs, own reward/done), three constructions
"""
import importlib.util
import io
import os
import sys
import types
import warnings
import zipfile

import numpy as np

REF = os.environ.get("QUADSIM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

sys.dont_write_bytecode = True
warnings.simplefilter("ignore", RuntimeWarning)


def load_reference():
    sys.path.insert(0, REF)
    ev = types.ModuleType("evdev")
    for n in ("InputDevice", "categorize", "ecodes", "KeyEvent"):
        setattr(ev, n, None)
    sys.modules["evdev"] = ev

    gym = types.ModuleType("gym")

    class Env:
        metadata = {}

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
            self.low, self.high, self.dtype = low, high, dtype
            self.shape = shape if shape is not None else np.asarray(low).shape

    gym.Env = Env
    spaces = types.ModuleType("gym.spaces"); spaces.Box = Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    utils.seeding = seeding
    error = types.ModuleType("gym.error"); logger = types.ModuleType("gym.logger")
    gym.spaces, gym.utils, gym.error, gym.logger = spaces, utils, error, logger
    for name, mod in (("gym", gym), ("gym.spaces", spaces), ("gym.utils", utils),
                      ("gym.utils.seeding", seeding), ("gym.error", error), ("gym.logger", logger)):
        sys.modules[name] = mod

    import dynamics.quadrotor as quadrotor
    import utils.transform as transform
    import controller.PIDController as pid

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    v0 = by_path("ref_docking_env", "gym-docking/gym_docking/envs/docking_env.py")
    v2 = by_path("ref_moving_docking_env", "gym-docking/gym_docking/envs/moving_docking_env.py")
    return quadrotor, transform, pid, v0, v2


quadrotor, transform, pid, env_v0, env_v2 = load_reference()
from scipy.integrate import RK45  # noqa: E402  (the reference's own integrator)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-22s %8.1f KiB  %s" % (name, os.path.getsize(path) / 1024.0,
                                     {k: v.shape for k, v in arrays.items()}))


# ----------------------------------------------------------------------------
def set_drone(drone, state, u_prev):
    """start the reference Drone from an arbitrary (state, u_prev): rebuild the
    integrator exactly as quadrotor.py:142 does so RK45's cached f0 is consistent."""
    drone.state = np.array(state, dtype=np.float64)
    drone.u = np.array(u_prev, dtype=np.float64)
    drone.integrator = RK45(drone.f, drone.t, drone.state, drone.t + drone.dt)


def limiter_margin(q):
    e = transform.quat2euler(q)
    lim = np.array([transform.deg2rad(85), transform.deg2rad(85), transform.deg2rad(175)])
    return float(np.min(np.abs(np.abs(e) - lim)))


def gen_g1(n_plain=1800, n_limit=700, seed=101):
    rs = np.random.RandomState(seed)
    drone = quadrotor.Drone()
    mg = drone.mass * drone.gravity
    S, UP, U, S2, UP2, FLAG, MARGIN = [], [], [], [], [], [], []

    def rand_case(want_limit):
        while True:
            s = np.zeros(13)
            s[0:3] = rs.uniform(-60, 60, 3)
            s[3:6] = rs.normal(0, 3, 3)
            if want_limit:
                # attitudes near/over the clamp: build from large euler angles
                e = rs.uniform(-1, 1, 3) * np.array([np.pi / 2 * 1.1, np.pi / 2 * 1.1, np.pi])
                k = rs.randint(3)
                lim = [1.4835298641951802, 1.4835298641951802, 3.0543261909900763][k]
                e[k] = rs.choice([-1, 1]) * (lim + rs.uniform(-0.02, 0.08))
                q = transform.euler2quat(e)
            else:
                q = rs.normal(0, 1, 4)
                q /= np.linalg.norm(q)
            s[6:10] = q * rs.uniform(0.9, 1.1)
            s[10:13] = rs.normal(0, 4, 3)
            up = np.array([rs.uniform(0, 4 * mg), *rs.normal(0, 6, 3)])
            if rs.rand() < 0.5:     # gentle command: rotor clamp mostly inactive
                u = np.array([rs.uniform(0.6 * mg, 3.0 * mg), *rs.normal(0, 0.01, 2), rs.normal(0, 0.02)])
            else:                   # harsh command: clamps active
                u = np.array([rs.uniform(-1.0, 5 * mg), *rs.normal(0, 0.25, 2), rs.normal(0, 0.02)])
            set_drone(drone, s, up)
            pre = s + drone.dt * drone.df(s, up)       # what RK45 will return (frozen RHS)
            margin = limiter_margin(pre[6:10])
            if margin < 1e-6:
                continue                                # knife-edge: the reference itself is ulp-decided
            s2 = np.array(drone.step(u.copy()), dtype=np.float64)
            fired = bool(np.all(s2[10:13] == 0.0) and np.max(np.abs(s2[6:10] - pre[6:10])) > 0)
            if want_limit and not fired:
                continue
            assert np.max(np.abs(s2[0:6] - pre[0:6])) < 1e-12, "RK45 != Euler"
            return s, up, u, s2, drone.u.copy(), fired, margin

    for i in range(n_plain + n_limit):
        c = rand_case(i >= n_plain)
        for lst, v in zip((S, UP, U, S2, UP2, FLAG, MARGIN), c):
            lst.append(v)
    FLAG = np.array(FLAG, np.uint8)
    print("g1: limiter hits %d / %d, clamps active in %d" % (
        FLAG.sum(), len(FLAG), int(np.sum(np.abs(np.array(UP2)[:, 0] - np.array(U)[:, 0]) > 1e-9))))
    save("g1_drone_step", state=np.array(S), u_prev=np.array(UP), u=np.array(U), state_out=np.array(S2),
         u_prev_out=np.array(UP2), limited=FLAG, margin=np.array(MARGIN), dt=np.array(drone.dt))


def gen_g2(n=400, seed=202):
    rs = np.random.RandomState(seed)
    q = rs.normal(0, 1, (n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[n // 2:] *= rs.uniform(0.6, 1.6, (n - n // 2, 1))      # un-normalised: reaches the r12 saturation branches
    # forced saturation: quats with 2(wx+yz) beyond +-1
    q[-8:] = np.array([[1, 1, 0, 0], [1, -1, 0, 0], [0.9, 0.9, 0.1, 0.05], [0.9, -0.9, 0.1, -0.05],
                       [0, 0, 1, 1], [0, 0, 1, -1], [1.2, 0.5, 0.1, 0.0], [0.8, -0.7, 0.0, 0.1]], float)
    e_in = rs.uniform(-np.pi, np.pi, (n, 3))
    q2e = np.array([transform.quat2euler(x) for x in q])
    e2q = np.array([transform.euler2quat(x) for x in e_in])
    q2r = np.array([transform.quat2rot(x) for x in q])
    q2r_drone = np.array([quadrotor.Drone.quat2rot(x) for x in q])
    assert np.array_equal(q2r, q2r_drone)
    # rot2euler on what state2rel feeds it: R_B @ R_A^T of two quat2rot outputs, plus raw matrices
    Rm = np.array([transform.quat2rot(q[i]) @ transform.quat2rot(q[(i * 7 + 3) % n]).T for i in range(n)])
    Rm[-6:, 1, 2] = np.array([1.0, 1.5, -1.0, -1.5, 0.999999, -0.999999])
    r2e = np.array([transform.rot2euler(x) for x in Rm])
    sat = int(np.sum(np.abs(Rm[:, 1, 2]) >= 1)), int(np.sum(np.abs(2 * (q[:, 0] * q[:, 1] + q[:, 2] * q[:, 3])) >= 1))
    print("g2: saturated rot2euler cases %d, quat2euler cases %d" % sat)
    save("g2_transforms", quat=q, quat2euler=q2e, euler=e_in, euler2quat=e2q, quat2rot=q2r, rot=Rm, rot2euler=r2e)


def gen_g12(n=300, seed=1212):
    """Drone.get_dock_port_state (dynamics/quadrotor.py:213-224) for random states and the three ports in use"""
    rs = np.random.RandomState(seed)
    S = np.zeros((n, 13))
    S[:, 0:3] = rs.normal(0, 20, (n, 3)); S[:, 3:6] = rs.normal(0, 2, (n, 3))
    q = rs.normal(0, 1, (n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[n // 2:] *= rs.uniform(0.7, 1.4, (n - n // 2, 1))
    S[:, 6:10] = q; S[:, 10:13] = rs.normal(0, 2, (n, 3))
    ports = np.array([[0.1, 0, 0], [-0.1, 0, 0], [0.05, 0, 0]])
    P = ports[rs.randint(0, 3, n)]
    pos, vel, quat, rate = [], [], [], []
    d = quadrotor.Drone()
    for i in range(n):
        d.reset(reset_state=S[i].copy(), dock_port=P[i].copy())
        dp = d.get_dock_port_state()
        pos.append(dp["pos"]); vel.append(dp["vel"]); quat.append(dp["quat"]); rate.append(dp["angular_rate"])
    save("g12_dock_port", state=S, port=P, pos=np.array(pos), vel=np.array(vel), quat=np.array(quat, np.float64),
         angular_rate=np.array(rate))


def gen_g3(n=400, seed=303):
    rs = np.random.RandomState(seed)
    ctl = pid.controller(0.086, 0.18)
    SD, SN, SL, U_PID, SD_PID, U_VEL, SD_VEL = [], [], [], [], [], [], []
    for i in range(n):
        sd = np.zeros(13)
        sd[0:3] = rs.uniform(-20, 20, 3); sd[3:6] = rs.normal(0, 0.5, 3)
        qd = transform.euler2quat(rs.uniform(-0.5, 0.5, 3) * np.array([1, 1, 6]))
        sd[6:10] = qd
        sd[12] = rs.normal(0, 0.2) if i % 4 == 0 else 0.0
        sn = np.zeros(13)
        sn[0:3] = sd[0:3] + rs.normal(0, 1.0, 3); sn[3:6] = rs.normal(0, 1, 3)
        qn = transform.euler2quat(rs.uniform(-0.7, 0.7, 3) * np.array([1, 1, 4]))
        sn[6:10] = qn * rs.uniform(0.97, 1.03)
        sn[10:13] = rs.normal(0, 1, 3)
        sl = sn.copy(); sl[3:6] += rs.normal(0, 0.1, 3)
        a = sd.copy(); u1 = ctl.PID(a, sn)
        b = sd.copy(); u2 = ctl.vel_controller(b, sn, sl)
        for lst, v in zip((SD, SN, SL, U_PID, SD_PID, U_VEL, SD_VEL), (sd, sn, sl, u1, a, u2, b)):
            lst.append(np.array(v))
    save("g3_controller", state_des=np.array(SD), state_now=np.array(SN), state_last=np.array(SL),
         u_pid=np.array(U_PID), state_des_after_pid=np.array(SD_PID),
         u_vel=np.array(U_VEL), state_des_after_vel=np.array(SD_VEL), mass=np.array(0.18))


# ----------------------------------------------------------------------------
def snapshot(env):
    rec = np.zeros(40)
    rec[0:13] = env.state_chaser
    rec[13:26] = env.state_target
    rec[26:30] = env.chaser.u
    rec[30:34] = env.target.u
    rec[34:38] = env.target_state_des[6:10]
    rec[38] = env.last_shaping
    rec[39] = env.t
    return rec


def run_traj(env, actions):
    """SB2-worker semantics: step; on done keep the terminal obs and reset()."""
    T = len(actions)
    rec_before = np.zeros((T, 40)); rec_after = np.zeros((T, 40))
    obs = np.zeros((T, 12)); rew = np.zeros(T); done = np.zeros(T, np.uint8); flags = np.zeros(T, np.uint8)
    reset_obs = np.full((T, 12), np.nan)
    first = env.reset()
    for t in range(T):
        rec_before[t] = snapshot(env)
        o, r, d, info = env.step(np.asarray(actions[t], dtype=np.float64))
        rec_after[t] = snapshot(env)
        obs[t], rew[t], done[t] = o, r, d
        flags[t] = (1 if info["flag_docking"] else 0) | (2 if info["done_overlimit"] else 0) | (4 if env.t >= 600 else 0)
        assert info["chaser"] is env.state_chaser and info["target"] is env.state_target
        if d:
            reset_obs[t] = env.reset()
    return dict(first_obs=np.array(first), rec_before=rec_before, rec_after=rec_after, obs=obs, reward=rew,
                done=done, flags=flags, reset_obs=reset_obs)


def mixed_actions(rs, T, block=500):
    a = np.zeros((T, 4), np.float32)
    for b0 in range(0, T, block):
        n = min(block, T - b0)
        if (b0 // block) % 2 == 0:
            a[b0:b0 + n] = rs.uniform(-1, 1, (n, 4))
        else:
            a[b0:b0 + n] = np.clip(-0.5 + 0.05 * rs.randn(n, 4), -1, 1)
    return a


def gen_g4(T=1500):
    for name, mod, cls, seed in (("g4_traj_v0", env_v0, "DockingEnv", 0), ("g4_traj_v2", env_v2, "MovingDockingEnv", 1)):
        rs = np.random.RandomState(seed)
        actions = mixed_actions(rs, T)
        env = getattr(mod, cls)()
        out = run_traj(env, actions)
        print("%s: episodes %d, docked steps %d, overtime %d" % (
            name, int(out["done"].sum()), int((out["flags"] & 1).sum()), int((out["flags"] & 4 > 0).sum())))
        save(name, actions=actions, **out)


def gen_g5():
    z = zipfile.ZipFile(os.path.join(REF, "trained_model", "best_model_v0.zip"))
    P = np.load(io.BytesIO(z.read("parameters")), allow_pickle=False)
    W = {k: P[k] for k in P.files}

    def policy(o):
        h = np.maximum(o.astype(np.float32) @ W["model/shared_fc0/w:0"] + W["model/shared_fc0/b:0"], 0)
        h = np.maximum(h @ W["model/pi_fc0/w:0"] + W["model/pi_fc0/b:0"], 0)
        return np.clip(h @ W["model/pi/w:0"] + W["model/pi/b:0"], -1, 1).astype(np.float32)

    env = env_v0.DockingEnv()
    o = env.reset()
    A, RB, RA, O, R, D, F = [], [], [], [], [], [], []
    for t in range(600):
        a = policy(np.asarray(o))
        RB.append(snapshot(env))
        o, r, d, info = env.step(a.astype(np.float64))
        RA.append(snapshot(env)); A.append(a); O.append(np.array(o)); R.append(r); D.append(d)
        F.append((1 if info["flag_docking"] else 0) | (2 if info["done_overlimit"] else 0) | (4 if env.t >= 600 else 0))
        if d:
            break
    # the policy's weights are data (a float32 MLP 12 -> 128 -> 128 -> 4, plus the value branch vf_fc0 -> vf and the
    # Gaussian's logstd that PPO2's Runner uses): kept as a fixture so the closed loop can be re-run on the GPU box,
    # where the reference tree does not exist
    save("policy_best_model_v0", w0=W["model/shared_fc0/w:0"], b0=W["model/shared_fc0/b:0"],
         w1=W["model/pi_fc0/w:0"], b1=W["model/pi_fc0/b:0"], w2=W["model/pi/w:0"], b2=W["model/pi/b:0"],
         wv1=W["model/vf_fc0/w:0"], bv1=W["model/vf_fc0/b:0"], wv2=W["model/vf/w:0"], bv2=W["model/vf/b:0"],
         logstd=W["model/pi/logstd:0"].reshape(-1))
    print("g5: steps %d, return %.4f, docked steps %d, last flags %d" % (
        len(A), float(np.sum(R)), int(np.sum(np.array(F) & 1)), F[-1]))
    save("g5_policy_episode", actions=np.array(A, np.float32), rec_before=np.array(RB), rec_after=np.array(RA),
         obs=np.array(O), reward=np.array(R), done=np.array(D, np.uint8), flags=np.array(F, np.uint8))


def gen_g6(T=2000):
    d2r = transform.deg2rad
    ini_state = np.zeros(13)
    ini_state[6:10] = transform.euler2quat(np.array([d2r(0.0), d2r(0.0), d2r(0.0)]))
    state_des = np.zeros(13)
    state_des[0:3] = np.array([-0.2, 0.2, 0.2])
    state_des[6:10] = transform.euler2quat(np.array([0.0, 0.0, 0.0]))
    sd0 = state_des.copy()
    quad = quadrotor.Drone()
    quad.reset(ini_state.copy())
    ctl = pid.controller(quad.get_arm_length(), quad.get_mass())
    S = np.zeros((T, 13)); U = np.zeros((T, 4)); TT = np.zeros(T)
    for t in range(T):
        s = quad.get_state()
        u = ctl.PID(state_des, s)
        U[t], S[t], TT[t] = u, s, quad.get_time()
        quad.step(u)
    print("g6: final state", np.round(quad.get_state()[:3], 4), "t_final", quad.get_time())
    save("g6_sim_pid", ini_state=ini_state, state_des=sd0, states=S, u=U, time=TT,
         final_state=np.array(quad.get_state()), final_state_des=state_des, t_final=np.array(quad.get_time()))


def gen_g7(T=400):
    out = {}
    sets = [(0.8, (0.85, 1.1, 1.2)), (1.2, (1.15, 0.9, 0.8)), (0.93, (1.0, 1.0, 1.0))]
    for kind, (mod, cls) in enumerate(((env_v0, "DockingEnv"), (env_v2, "MovingDockingEnv"))):
        for j, (ms, iscale) in enumerate(sets):
            env = getattr(mod, cls)()
            m = 0.18 * ms
            I = np.diag([0.00025 * iscale[0], 0.000232 * iscale[1], 0.0003738 * iscale[2]])
            for d in (env.chaser, env.target):
                d.mass = m; d.Inertia = I.copy(); d.F_max = 4 * m * d.gravity
            env.target_controller.mass = m
            env.action_mean = np.ones(4) * m * env.chaser.gravity / 2.0
            env.action_std = np.ones(4) * m * env.chaser.gravity / 2.0
            rs = np.random.RandomState(700 + 10 * kind + j)
            actions = mixed_actions(rs, T, block=200)
            tr = run_traj(env, actions)      # run_traj resets first -> RK45 caches a consistent f0
            key = "k%d_s%d_" % (kind, j)
            out[key + "par"] = np.array([m, I[0, 0], I[1, 1], I[2, 2]])
            out[key + "actions"] = actions
            for k in ("rec_before", "rec_after", "obs", "reward", "done", "flags"):
                out[key + k] = tr[k]
    save("g7_domain_rand", **out)


def gen_g8(T=900):
    """docking-v1 (imitating_docking_env.py): v0 + chaser position jitter drawn ONCE at construction"""
    v1 = importlib.util.spec_from_file_location(
        "ref_imitating_docking_env", os.path.join(REF, "gym-docking/gym_docking/envs/imitating_docking_env.py"))
    mod = importlib.util.module_from_spec(v1); v1.loader.exec_module(mod)
    out = {}
    for j, seed in enumerate((11, 12, 13)):
        np.random.seed(seed)                    # the ctor uses the global numpy RNG (:34)
        env = mod.ImitatingDockingEnv()
        rs = np.random.RandomState(800 + j)
        actions = mixed_actions(rs, T, block=300)
        tr = run_traj(env, actions)
        key = "e%d_" % j
        out[key + "chaser_ini_state"] = np.array(env.chaser_ini_state)
        out[key + "target_ini_state"] = np.array(env.target_ini_state)
        out[key + "actions"] = actions
        for k in ("first_obs", "rec_before", "rec_after", "obs", "reward", "done", "flags", "reset_obs"):
            out[key + k] = tr[k]
        assert np.max(np.abs(env.chaser_ini_state[0:3] - [8, -50, 5])) <= 0.3
    print("g8: episodes", [int(out["e%d_done" % j].sum()) for j in range(3)])
    save("g8_traj_v1", **out)


def gen_g9(T=1200):
    """hovering-v0 (hovering_env.py): one drone, raw-state obs, own reward/done; ini_state drawn at construction"""
    hv = importlib.util.spec_from_file_location(
        "ref_hovering_env", os.path.join(REF, "gym-docking/gym_docking/envs/hovering_env.py"))
    mod = importlib.util.module_from_spec(hv); hv.loader.exec_module(mod)
    out = {}
    for j, seed in enumerate((21, 22, 23)):
        np.random.seed(seed)
        env = mod.HoveringEnv()
        rs = np.random.RandomState(900 + j)
        a = np.zeros((T, 4), np.float32)
        for b0 in range(0, T, 200):            # blocks: U(0,1), near-hover (0.25 = weight/4 per rotor), full thrust
            m = (b0 // 200) % 3
            a[b0:b0 + 200] = (rs.uniform(0, 1, (200, 4)) if m == 0 else
                              np.clip(0.25 + 0.02 * rs.randn(200, 4), 0, 1) if m == 1 else
                              np.clip(0.9 + 0.1 * rs.randn(200, 4), 0, 1))
        if j == 2:
            a[:] = np.clip(0.97 + 0.03 * rs.randn(T, 4), 0, 1)    # climbs away: reaches the done branch (|pos| > 100)
        s = np.array(env.reset(), dtype=np.float64)
        SB, UB, SA, UA, R, D = [], [], [], [], [], []
        for t in range(T):
            SB.append(np.array(env.drone.state, dtype=np.float64)); UB.append(np.array(env.drone.u))
            o, r, d, _ = env.step(a[t].astype(np.float64))
            SA.append(np.array(o, dtype=np.float64)); UA.append(np.array(env.drone.u)); R.append(r); D.append(d)
            if d:
                env.reset()
        key = "e%d_" % j
        out[key + "ini_state"] = np.array(env.ini_state)
        out[key + "actions"] = a
        out[key + "state_before"] = np.array(SB); out[key + "u_before"] = np.array(UB)
        out[key + "state_after"] = np.array(SA); out[key + "u_after"] = np.array(UA)
        out[key + "reward"] = np.array(R); out[key + "done"] = np.array(D, np.uint8)
    # crafted single steps around the +1 bonus ball (|pos err| < 0.1 and |vel| < 0.1, hovering_env.py:63)
    np.random.seed(24)
    env = mod.HoveringEnv()
    rs = np.random.RandomState(950)
    SB, UB, A, SA, UA, R, D, M = [], [], [], [], [], [], [], []
    while len(SB) < 300:
        s = np.zeros(13)
        s[0:3] = np.array([0, 0, 5.0]) + rs.normal(0, 0.06, 3)
        s[3:6] = rs.normal(0, 0.06, 3)
        s[6:10] = transform.euler2quat(rs.uniform(-0.3, 0.3, 3))
        s[10:13] = rs.normal(0, 0.3, 3)
        up = np.array([0.18 * 9.81 + rs.normal(0, 0.2), *rs.normal(0, 2, 3)])
        a = np.clip(0.25 + 0.1 * rs.randn(4), 0, 1).astype(np.float32)
        set_drone(env.drone, s, up)
        o, r, d, _ = env.step(a.astype(np.float64))
        o = np.array(o, dtype=np.float64)
        margin = min(abs(np.linalg.norm(o[0:3] - [0, 0, 5]) - 0.1), abs(np.linalg.norm(o[3:6]) - 0.1))
        SB.append(s); UB.append(up); A.append(a); SA.append(o); UA.append(np.array(env.drone.u)); R.append(r); D.append(d); M.append(margin)
    out.update(c_state_before=np.array(SB), c_u_before=np.array(UB), c_actions=np.array(A), c_state_after=np.array(SA),
               c_u_after=np.array(UA), c_reward=np.array(R), c_done=np.array(D, np.uint8), c_margin=np.array(M))
    print("g9: crafted bonus steps", int((np.array(R) > 1.0).sum()), "of", len(R))
    print("g9: done counts", [int(out["e%d_done" % j].sum()) for j in range(3)],
          "bonus steps", [int((out["e%d_reward" % j] > 1.0).sum()) for j in range(3)])
    save("g9_hovering", **out)


def gen_g11(total_step=900):
    """PID expert on docking-v0: control flow transcribed from run_expert_policy.py:39-69 (the script itself needs
    stable_baselines / a TkAgg display); every number comes from the reference's env and controller classes."""
    env = env_v0.DockingEnv()
    obs = env.reset()
    control = pid.controller(env.chaser.get_arm_length(), env.chaser.get_mass())
    state_des = env.chaser_ini_state            # aliased, as in the script (:44)
    kp, kd = 0.35, 0
    info_lst, O, A, U, R, D, SD, SC, ST = [], [], [], [], [], [], [], [], []
    for t in range(total_step):
        obss = np.array(obs).flatten()
        state_last = info_lst[t - 1]["chaser"] if t != 0 else env.chaser_ini_state
        des_vel = kp * (env.state_target[0:3] + np.array([-0.2, 0, 0]) - env.state_chaser[0:3]) + kd * (-env.state_chaser[3:6])
        if t != 0:
            state_des[3:6] = des_vel
        SC.append(np.array(env.state_chaser)); ST.append(np.array(env.state_target))
        action = control.vel_controller(state_des, env.state_chaser, state_last)
        u = (np.linalg.inv(env.chaser.rotor2control) @ action - env.action_mean) / env.action_std
        SD.append(np.array(state_des))
        obs, reward, done, info = env.step(u)
        O.append(obss); A.append(np.array(u)); U.append(np.array(action)); R.append(reward); D.append(done)
        info_lst.append(info)
        if done:
            break
    print("g11: steps %d, return %.4f, docked steps %d, done %s" % (len(A), float(np.sum(R)),
          int(sum(i["flag_docking"] for i in info_lst)), D[-1]))
    save("g11_expert_episode", obs=np.array(O), actions=np.array(A), u=np.array(U), rewards=np.array(R),
         done=np.array(D, np.uint8), state_des_after=np.array(SD), chaser=np.array(SC), target=np.array(ST),
         last_obs=np.array(obs), kp_kd=np.array([kp, kd]))


def _ref_source_block(path, start_marker, end_marker):
    """lines [start_marker .. end_marker] of a reference file, dedented -- executed, never stored"""
    import textwrap
    lines = open(os.path.join(REF, path)).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if start_marker in l)
    i1 = next(i for i, l in enumerate(lines) if end_marker in l and i >= i0)
    return textwrap.dedent("\n".join(lines[i0:i1 + 1]))


def gen_g10():
    """GAE(lambda) + swap_and_flatten of the in-tree PPO2 Runner, rl_baselines/ppo2/ppo2.py:507-520,531-539.
    The module itself needs TensorFlow (absent), so the generator EXECUTES those source lines of the reference
    file as they stand, in a namespace holding the Runner's local variables."""
    gae_src = _ref_source_block("rl_baselines/ppo2/ppo2.py", "mb_advs = np.zeros_like(mb_rewards)", "mb_returns = mb_advs + mb_values")
    flat_src = _ref_source_block("rl_baselines/ppo2/ppo2.py", "def swap_and_flatten(arr):", "return arr.swapaxes(0, 1)")
    ns_f = {"np": np}
    exec(compile(flat_src, "ppo2.py:swap_and_flatten", "exec"), ns_f)
    out = {}
    for j, (T, N, gamma, lam, seed) in enumerate(((600, 10, 0.99, 0.95, 1), (128, 37, 0.99, 0.95, 2), (64, 130, 0.9, 1.0, 3), (5, 3, 1.0, 0.0, 4))):
        rs = np.random.RandomState(seed)
        mb_rewards = rs.normal(0, 1, (T, N)).astype(np.float32)
        mb_values = rs.normal(0, 2, (T, N)).astype(np.float32)
        mb_dones = rs.rand(T, N) < (0.03 if T > 10 else 0.3)
        last_values = rs.normal(0, 2, N).astype(np.float32)
        last_dones = rs.rand(N) < 0.2
        self = types.SimpleNamespace(n_steps=T, dones=last_dones, gamma=gamma, lam=lam)
        ns = dict(np=np, self=self, mb_rewards=mb_rewards, mb_values=mb_values, mb_dones=mb_dones, last_values=last_values)
        exec(compile(gae_src, "ppo2.py:gae", "exec"), ns)
        key = "c%d_" % j
        out.update({key + "rewards": mb_rewards, key + "values": mb_values, key + "dones": mb_dones.astype(np.uint8),
                    key + "last_values": last_values, key + "last_dones": last_dones.astype(np.uint8),
                    key + "gamma_lam": np.array([gamma, lam]), key + "advs": ns["mb_advs"], key + "returns": ns["mb_returns"],
                    key + "flat_returns": ns_f["swap_and_flatten"](ns["mb_returns"])})
        if T <= 64:
            obs = rs.normal(0, 1, (T, N, 12)).astype(np.float32)
            out[key + "obs"] = obs
            out[key + "flat_obs"] = ns_f["swap_and_flatten"](obs)
        assert ns["mb_advs"].dtype == np.float32
    save("g10_gae", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12"]
    for w in which:
        globals()["gen_" + w]()
