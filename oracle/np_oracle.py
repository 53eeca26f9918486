"""NumPy twin of the CPU oracle: the closed form of one ``env.step``, vectorised over N envs
in float64.

TEST INFRASTRUCTURE ONLY (same status as quadsim_oracle.c): imported by tests/ and bench.py's CPU legs, never by the
product package.  It restates the reference independently of the C restatement -- array code instead of scalar C -- so
the two can be held against each other and against the reference's fixtures (tests/test_oracle_vs_golden.py).

Reference lines followed: dynamics/quadrotor.py:80-113 (df), :126-144 (step; RK45 over the frozen RHS == Euler with the
previous limited control), :146-168 (attitude_limit), :171-187 (u_limit), :213-224 (dock port), :226-245 (quat2rot);
utils/transform.py:23-46 (rot2euler), :94-120 (quat2euler), :123-136 (euler2quat); controller/PIDController.py:52-141;
gym-docking/gym_docking/envs/docking_env.py:104-231, :257-295; moving_docking_env.py:111-192.
Record layout = the C oracle's: [N,40] = chaser 13 | target 13 | u_chaser 4 | u_target 4 | q_des 4 | last_shaping | t.
"""
import numpy as np

G = 9.81
L = 0.086
LAM = 1.5e-9 / 6.11e-8
LIM85, LIM175, LIM10 = np.deg2rad(85.0), np.deg2rad(175.0), np.deg2rad(10.0)
PAR_NOMINAL = (0.18, 0.00025, 0.000232, 0.0003738)


def quat2rot(q):
    """quadrotor.py:226-245 / transform.py:4-20: element-wise qa_hat*qa_hat with the NORMALISED vector part and the
    UN-normalised scalar part; unit diagonal.  q [N,4] -> R [N,3,3]"""
    n = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, n1, n2, n3 = q[:, 0], n[:, 1], n[:, 2], n[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = R[:, 1, 1] = R[:, 2, 2] = 1.0
    R[:, 0, 1] = 2 * n3 * n3 - 2 * w * n3
    R[:, 1, 0] = 2 * n3 * n3 + 2 * w * n3
    R[:, 0, 2] = 2 * n2 * n2 + 2 * w * n2
    R[:, 2, 0] = 2 * n2 * n2 - 2 * w * n2
    R[:, 1, 2] = 2 * n1 * n1 - 2 * w * n1
    R[:, 2, 1] = 2 * n1 * n1 + 2 * w * n1
    return R


def _euler_from(r10, r11, r12, r02, r22):
    phi = np.arcsin(np.clip(r12, -1.0, 1.0))
    sat = (r12 >= 1.0) | (r12 < -1.0)
    theta = np.where(sat, 0.0, np.arctan2(-r02, r22))
    psi = np.arctan2(-r10, r11)
    return phi, theta, psi


def quat2euler(q):
    """transform.py:94-120 -> (roll, pitch, yaw)"""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return _euler_from(2 * (x * y - w * z), w * w - x * x + y * y - z * z, 2 * (w * x + y * z), 2 * (x * z - w * y),
                       w * w - x * x - y * y + z * z)


def rot2euler(R):
    """transform.py:23-46"""
    return _euler_from(R[:, 1, 0], R[:, 1, 1], R[:, 1, 2], R[:, 0, 2], R[:, 2, 2])


def euler2quat(r, p, y):
    """transform.py:123-136"""
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    return np.stack([cr * cp * cy - sr * sp * sy, sr * cp * cy - cr * sp * sy, sr * cp * sy + cr * sp * cy,
                     cr * cp * sy + sr * sp * cy], 1)


def drone_df(s, u, par):
    """quadrotor.py:80-113; par [N,4] = mass, Ixx, Iyy, Izz"""
    q, w = s[:, 6:10], s[:, 10:13]
    R = quat2rot(q)
    ds = np.empty_like(s)
    ds[:, 0:3] = s[:, 3:6]
    Fm = u[:, 0] / par[:, 0]
    ds[:, 3] = R[:, 2, 0] * Fm
    ds[:, 4] = R[:, 2, 1] * Fm
    ds[:, 5] = Fm - G
    kq = 2.0 * (1.0 - np.sum(q * q, axis=1))
    k = np.stack([-w[:, 0] * q[:, 1] - w[:, 1] * q[:, 2] - w[:, 2] * q[:, 3],
                  w[:, 0] * q[:, 0] - w[:, 1] * q[:, 2] + w[:, 2] * q[:, 3],
                  w[:, 1] * q[:, 0] + w[:, 2] * q[:, 1] - w[:, 0] * q[:, 3],
                  w[:, 2] * q[:, 0] - w[:, 1] * q[:, 1] + w[:, 0] * q[:, 2]], 1)
    ds[:, 6:10] = -0.5 * k + kq[:, None] * q
    I = par[:, 1:4]
    ds[:, 10:13] = (I * u[:, 1:4] - np.cross(w, I * w)) / I
    return ds


def attitude_limit(s):
    """quadrotor.py:146-168 + write-back :135-138: sequential overriding ifs, >= / <= tie"""
    r, p, y = quat2euler(s[:, 6:10])
    lim = np.zeros(len(s), bool)
    q = s[:, 6:10].copy()
    a = np.abs(r) >= LIM85
    q[a] = euler2quat(np.sign(r[a]) * LIM85, p[a], y[a]); lim |= a
    b = np.abs(p) >= LIM85
    q[b] = euler2quat(r[b], np.sign(p[b]) * LIM85, y[b]); lim |= b
    c = np.abs(y) >= LIM175
    q[c] = euler2quat(r[c], p[c], np.sign(y[c]) * LIM175); lim |= c
    inside = (np.abs(r) <= LIM85) & (np.abs(p) <= LIM85) & (np.abs(y) <= LIM175)
    lim &= ~inside
    s = s.copy()
    s[lim, 6:10] = q[lim]
    s[lim, 10:13] = 0.0
    return s, lim


def u_limit(u, par):
    """quadrotor.py:171-187: per-rotor clamp [0, m g]; yaw moment passes through"""
    mg = par[:, 0] * G
    f4, a = u[:, 0] / 4.0, 0.5 / L
    p0 = np.clip(f4 - a * u[:, 2], 0.0, mg); p1 = np.clip(f4 + a * u[:, 1], 0.0, mg)
    p2 = np.clip(f4 + a * u[:, 2], 0.0, mg); p3 = np.clip(f4 - a * u[:, 1], 0.0, mg)
    return np.stack([p0 + p1 + p2 + p3, L * (p1 - p3), L * (p2 - p0), u[:, 3]], 1)


def drone_step(s, u_prev, u, par, dt=0.02):
    """quadrotor.py:126-144 -> (state', limited control to store, limiter flag)"""
    s2, lim = attitude_limit(s + dt * drone_df(s, u_prev, par))
    return s2, u_limit(u, par), lim


def target_control(kind, s, qdes, par, vdes_x):
    """controller.PID (kind 0, PIDController.py:76-104,:179-185) / vel_controller (kind 1, :106-141) on the target;
    rewrites q_des; dv = 0 inside the envs (moving_docking_env.py:117)"""
    if kind == 0:
        ax = -1.0 * (10.0 - s[:, 0]) - 1.65 * (0.0 - s[:, 3])
        ay = -1.0 * (-50.0 - s[:, 1]) - 1.65 * (0.0 - s[:, 4])
        az = 50.0 * (5.0 - s[:, 2]) + 8.0 * (0.0 - s[:, 5])
    else:
        ax = -0.7 * (vdes_x - s[:, 3])
        ay = -0.7 * (0.0 - s[:, 4])
        az = 1.0 * (0.0 - s[:, 5])
    m = par[:, 0]
    F = m * G + m * az
    psi = quat2euler(qdes)[2]
    phi_d = (ax * np.sin(psi) - ay * np.cos(psi)) / G
    th_d = (ax * np.cos(psi) + ay * np.sin(psi)) / G
    qdes = euler2quat(phi_d, th_d, psi)
    dr, dp, dy = quat2euler(qdes)
    nr, np_, ny = quat2euler(s[:, 6:10])
    M = np.stack([-10.0 * (dr - nr) + 5.1 * (0.0 - s[:, 10]), -10.0 * (dp - np_) + 5.1 * (0.0 - s[:, 11]),
                  -9.5 * (dy - ny) + 4.0 * (0.0 - s[:, 12])], 1)
    return np.concatenate([F[:, None], M], 1), qdes


def rel_obs(sc, st):
    """dock ports (quadrotor.py:213-224) + state2rel (docking_env.py:257-295)"""
    RA, RB = quat2rot(sc[:, 6:10]), quat2rot(st[:, 6:10])
    port_c, port_t = np.array([0.1, 0.0, 0.0]), np.array([-0.1, 0.0, 0.0])
    bc = np.einsum("nji,j->ni", RA, port_c)
    bt = np.einsum("nji,j->ni", RB, port_t)
    pos = (st[:, 0:3] + bt) - (sc[:, 0:3] + bc)
    vel = (st[:, 3:6] + np.cross(st[:, 10:13], bt)) - (sc[:, 3:6] + np.cross(sc[:, 10:13], bc))
    R = np.einsum("nij,nkj->nik", RB, RA)
    phi, theta, psi = rot2euler(R)
    w = np.einsum("nij,nj->ni", RB, st[:, 10:13]) - np.einsum("nij,nj->ni", np.einsum("nij,njk->nik", R, RA), sc[:, 10:13])
    P, Q, Rr = w[:, 0], w[:, 1], w[:, 2]
    k = Rr * np.cos(theta) - P * np.sin(theta)
    rates = np.stack([P * np.cos(theta) + Rr * np.sin(theta), Q - np.tan(phi) * k, k / np.cos(phi)], 1)
    return np.concatenate([pos, vel, np.stack([phi, theta, psi], 1), rates], 1)


def env_step(rec, actions, par=None, kind=0, dt=0.02):
    """DockingEnv.step (kind 0) / MovingDockingEnv.step (kind 1) for N envs, no reset.
    rec [N,40] float64 (not modified), actions [N,4] -> (rec', obs [N,12], reward [N], done [N] bool, flags [N] uint8)"""
    rec = np.array(rec, np.float64)
    a = np.asarray(actions, np.float64)
    n = len(rec)
    par = np.tile(np.array(PAR_NOMINAL), (n, 1)) if par is None else np.asarray(par, np.float64).reshape(n, 4)
    sc, st, uc, ut, qd = rec[:, 0:13], rec[:, 13:26], rec[:, 26:30], rec[:, 30:34], rec[:, 34:38]
    t = rec[:, 39] + 1.0
    mean = 0.5 * par[:, 0:1] * G
    f = mean * a + mean                                              # docking_env.py:115 with :98-99
    u_c = np.stack([f.sum(1), L * (f[:, 1] - f[:, 3]), L * (f[:, 2] - f[:, 0]), LAM * (f[:, 0] - f[:, 1] + f[:, 2] - f[:, 3])], 1)
    rmax, vdes_x = (3.0, 0.0) if kind == 0 else (10.0, 0.2)
    u_t, qd2 = target_control(kind, st, qd, par, vdes_x)             # from the target state BEFORE stepping
    st2, ut2, lim_t = drone_step(st, ut, u_t, par, dt)
    sc2, uc2, lim_c = drone_step(sc, uc, u_c, par, dt)
    obs = rel_obs(sc2, st2)
    npos, nvel = np.linalg.norm(obs[:, 0:3], axis=1), np.linalg.norm(obs[:, 3:6], axis=1)
    docked = (npos < 0.1) & (nvel < 0.1) & (np.abs(obs[:, 6]) < LIM10) & (np.abs(obs[:, 7]) < LIM10) & (np.abs(obs[:, 8]) < LIM10)
    over = (npos >= rmax) | (sc2[:, 2] <= 0.1)
    overtime = t >= 600.0
    shaping = (-10.0 * npos / rmax - nvel - 10.0 * np.linalg.norm(obs[:, 6:9], axis=1) / np.pi
               - np.linalg.norm(obs[:, 9:12], axis=1) - 0.1 * np.linalg.norm(a, axis=1) + docked)
    reward = shaping - rec[:, 38]
    out = np.concatenate([sc2, st2, uc2, ut2, qd2, shaping[:, None], t[:, None]], 1)
    flags = (docked * 1 + over * 2 + overtime * 4 + lim_c * 8 + lim_t * 16).astype(np.uint8)
    return out, obs, reward, over | overtime, flags
