"""shared helpers of the GPU parity tests"""
import numpy as np

from oracle.pyoracle import PAR_NOMINAL, rec_pack

# Tolerances of the HIP fp32 path against the f64 oracle, per single step from
# identical (f32-rounded) inputs.  north_star: 1e-5 relative.  State components
# and observations are compared element-wise with rtol 1e-5 plus an absolute
# floor of 1e-5 x the natural scale of the block (1 for unit quaternions and
# O(1) rates; positions ~50 m so rtol dominates there).  The observation's
# relative position is a difference of two ~50 m positions: its absolute floor
# is 1e-5 x 2 m.  Reward = shaping - last_shaping with |shaping| ~ 2..10, so
# its bound is 1e-5 x max(1, |shaping|) x 2.
STATE_TOL = dict(rtol=1e-5, atol=1e-5)
OBS_TOL = dict(rtol=1e-5, atol=2e-5)


def reward_atol(shaping):
    return 2e-5 * np.maximum(1.0, np.abs(shaping))


def state_to_rec(st, dtype=np.float64):
    return rec_pack(st["chaser"], st["target"], st["u_prev"][:, :4], st["u_prev"][:, 4:], st["qdes"],
                    st["last_shaping"], st["t"], dtype=dtype)


def set_env_from_rec(env, rec):
    rec = np.asarray(rec, np.float32)
    env.set_state(chaser=rec[:, 0:13], target=rec[:, 13:26], u_prev=rec[:, 26:34], qdes=rec[:, 34:38],
                  last_shaping=rec[:, 38], t=rec[:, 39])


def threshold_margin(obs, chaser_z, t, rmax):
    """distance of each env from the nearest done / docking decision threshold"""
    npos = np.linalg.norm(obs[:, 0:3], axis=1)
    nvel = np.linalg.norm(obs[:, 3:6], axis=1)
    m = np.minimum(np.abs(npos - rmax), np.abs(chaser_z - 0.1))
    m = np.minimum(m, np.abs(npos - 0.1))
    m = np.minimum(m, np.abs(nvel - 0.1))
    for j in (6, 7, 8):
        m = np.minimum(m, np.abs(np.abs(obs[:, j]) - np.deg2rad(10)))
    return m


def tile_par(n, par=PAR_NOMINAL, dtype=np.float64):
    return np.tile(np.asarray(par, dtype), (n, 1))
