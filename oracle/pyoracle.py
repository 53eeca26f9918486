"""ctypes front-end of the CPU oracle (oracle/quadsim_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of quadsim_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package quadsim_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QSO_LIB") or os.path.join(HERE, "libqso.so")   # override: the sanitizer build (make asan)

REC_LEN = 40
REC_SC, REC_ST, REC_UC, REC_UT, REC_QD, REC_LS, REC_T = 0, 13, 26, 30, 34, 38, 39
STREAM_AUTORESET, STREAM_RESET, STREAM_ACTIONS = 0, 1, 2

# nominal per-env params (mass, Ixx, Iyy, Izz): dynamics/quadrotor.py:16-19
PAR_NOMINAL = (0.18, 0.00025, 0.000232, 0.0003738)
# no randomisation: zero half-ranges, unit scales
RR_NONE = (0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0)


def build(force=False):
    """Compile libqso.so with gcc (make -C oracle)."""
    src = os.path.join(HERE, "quadsim_oracle.c")
    if os.environ.get("QSO_LIB"):
        return LIB_PATH
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= os.path.getmtime(src)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """One precision instantiation of the oracle ('f64' is THE oracle)."""

    def __init__(self, prec="f64"):
        assert prec in ("f64", "f32")
        self.prec = prec
        self.dtype = np.float64 if prec == "f64" else np.float32
        self.creal = C.c_double if prec == "f64" else C.c_float
        self.L = lib()

    def _f(self, name):
        return getattr(self.L, "%s_%s" % (name, self.prec))

    def _a(self, x, shape=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=self.dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    # ---- transforms -------------------------------------------------------
    def quat2rot(self, q):
        q = self._a(q, (4,)); R = np.zeros(9, self.dtype)
        self._f("qso_quat2rot")(_p(q), _p(R)); return R.reshape(3, 3)

    def quat2euler(self, q):
        q = self._a(q, (4,)); e = np.zeros(3, self.dtype)
        self._f("qso_quat2euler")(_p(q), _p(e)); return e

    def euler2quat(self, e):
        e = self._a(e, (3,)); q = np.zeros(4, self.dtype)
        self._f("qso_euler2quat")(_p(e), _p(q)); return q

    def rot2euler(self, R):
        R = self._a(R, (9,)); e = np.zeros(3, self.dtype)
        self._f("qso_rot2euler")(_p(R), _p(e)); return e

    # ---- drone --------------------------------------------------------------
    def drone_df(self, s, u, par=PAR_NOMINAL):
        s = self._a(s, (13,)); u = self._a(u, (4,)); par = self._a(par, (4,))
        ds = np.zeros(13, self.dtype)
        self._f("qso_drone_df")(_p(s), _p(u), _p(par), _p(ds)); return ds

    def u_limit(self, u, par=PAR_NOMINAL):
        u = self._a(u, (4,)); par = self._a(par, (4,)); out = np.zeros(4, self.dtype)
        self._f("qso_u_limit")(_p(u), _p(par), _p(out)); return out

    def drone_step(self, s, u_prev, u, par=PAR_NOMINAL, dt=0.02, integ=0):
        """-> (state', u_prev', limiter_fired)"""
        s = self._a(s, (13,)).copy(); up = self._a(u_prev, (4,)).copy()
        u = self._a(u, (4,)); par = self._a(par, (4,))
        f = self._f("qso_drone_step"); f.restype = C.c_int
        over = f(_p(s), _p(up), _p(u), _p(par), self.creal(dt), C.c_int(integ))
        return s, up, int(over)

    def dock_port(self, s, port):
        s = self._a(s, (13,)); port = self._a(port, (3,))
        pos = np.zeros(3, self.dtype); vel = np.zeros(3, self.dtype)
        self._f("qso_dock_port")(_p(s), _p(port), _p(pos), _p(vel)); return pos, vel

    # ---- controller -----------------------------------------------------------
    def ctrl_pid(self, sdes, s, mass=0.18):
        """-> (u, mutated state_des)"""
        sdes = self._a(sdes, (13,)).copy(); s = self._a(s, (13,)); u = np.zeros(4, self.dtype)
        self._f("qso_ctrl_pid")(_p(sdes), _p(s), self.creal(mass), _p(u)); return u, sdes

    def ctrl_vel(self, sdes, s, s_last, mass=0.18):
        sdes = self._a(sdes, (13,)).copy(); s = self._a(s, (13,)); sl = self._a(s_last, (13,))
        u = np.zeros(4, self.dtype)
        self._f("qso_ctrl_vel")(_p(sdes), _p(s), _p(sl), self.creal(mass), _p(u)); return u, sdes

    # ---- env ----------------------------------------------------------------
    def rel_obs(self, sc, st):
        sc = self._a(sc, (13,)); st = self._a(st, (13,)); o = np.zeros(12, self.dtype)
        self._f("qso_rel_obs")(_p(sc), _p(st), _p(o)); return o

    def env_init(self, n=1):
        rec = np.zeros((n, REC_LEN), self.dtype)
        for i in range(n):
            self._f("qso_env_init")(_p(rec[i]))
        return rec

    def env_step(self, rec, a, par=PAR_NOMINAL, kind=0, dt=0.02, integ=0):
        """single env, no auto-reset: -> (rec', obs, reward, done, flags)"""
        rec = self._a(rec, (REC_LEN,)).copy(); a = self._a(a, (4,)); par = self._a(par, (4,))
        obs = np.zeros(12, self.dtype); rew = self.creal(0); done = C.c_int(0); flags = C.c_int(0)
        self._f("qso_env_step")(_p(rec), _p(a), _p(par), C.c_int(kind), self.creal(dt), C.c_int(integ),
                                _p(obs), C.byref(rew), C.byref(done), C.byref(flags))
        return rec, obs, float(rew.value), bool(done.value), int(flags.value)

    def vec_step(self, rec, par, actions, kind=0, dt=0.02, integ=0, auto_reset=True, randomise=0,
                 seed=0, step_idx=0, gid0=0, rr=RR_NONE, par_nom=PAR_NOMINAL, want_term=False):
        """in-place on rec/par ([N,40]/[N,4] arrays of this precision).
        -> (obs[N,12], reward[N], done[N] u8, flags[N] u8, term_obs or None)"""
        n = rec.shape[0]
        assert rec.dtype == self.dtype and rec.flags.c_contiguous and rec.shape == (n, REC_LEN)
        assert par.dtype == self.dtype and par.flags.c_contiguous and par.shape == (n, 4)
        actions = self._a(actions, (n, 4))
        obs = np.zeros((n, 12), self.dtype); rew = np.zeros(n, self.dtype)
        done = np.zeros(n, np.uint8); flags = np.zeros(n, np.uint8)
        term = np.full((n, 12), np.nan, self.dtype) if want_term else None
        rr = np.asarray(rr, np.float32); pn = np.asarray(par_nom, np.float32)
        self._f("qso_vec_step")(C.c_int64(n), _p(rec), _p(par), _p(actions), C.c_int(kind), self.creal(dt),
                                C.c_int(integ), C.c_int(int(auto_reset)), C.c_int(randomise),
                                C.c_uint64(seed), C.c_uint64(step_idx), C.c_uint64(gid0), _p(rr), _p(pn),
                                _p(obs), _p(rew), _p(done), _p(flags), _p(term))
        return obs, rew, done, flags, term

    def vec_reset(self, rec, par, mask=None, randomise=0, seed=0, step_idx=0, gid0=0,
                  rr=RR_NONE, par_nom=PAR_NOMINAL):
        n = rec.shape[0]
        assert rec.dtype == self.dtype and par.dtype == self.dtype
        obs = np.zeros((n, 12), self.dtype)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rr = np.asarray(rr, np.float32); pn = np.asarray(par_nom, np.float32)
        self._f("qso_vec_reset")(C.c_int64(n), _p(rec), _p(par), _p(m), C.c_int(randomise), C.c_uint64(seed),
                                 C.c_uint64(step_idx), C.c_uint64(gid0), _p(rr), _p(pn), _p(obs))
        return obs

    def vec_rollout(self, rec, par, actions, kind=0, dt=0.02, integ=0, randomise=0, seed=0,
                    step_idx0=0, gid0=0, rr=RR_NONE, par_nom=PAR_NOMINAL):
        """actions [T,N,4] -> obs[T,N,12], reward[T,N], done[T,N], flags[T,N]; rec/par in place."""
        actions = self._a(actions)
        T, n = actions.shape[0], actions.shape[1]
        obs = np.zeros((T, n, 12), self.dtype); rew = np.zeros((T, n), self.dtype)
        done = np.zeros((T, n), np.uint8); flags = np.zeros((T, n), np.uint8)
        rr = np.asarray(rr, np.float32); pn = np.asarray(par_nom, np.float32)
        self._f("qso_vec_rollout")(C.c_int64(T), C.c_int64(n), _p(rec), _p(par), _p(actions), C.c_int(kind),
                                   self.creal(dt), C.c_int(integ), C.c_int(randomise), C.c_uint64(seed),
                                   C.c_uint64(step_idx0), C.c_uint64(gid0), _p(rr), _p(pn),
                                   _p(obs), _p(rew), _p(done), _p(flags))
        return obs, rew, done, flags

    def vec_step_stored_init(self, rec, par, actions, init, kind=0, dt=0.02, integ=0, auto_reset=True,
                             want_term=False):
        """docking-v1 style: reset returns to init[N,26] (chaser_ini_state, target_ini_state)"""
        n = rec.shape[0]
        actions = self._a(actions, (n, 4)); init = self._a(init, (n, 26))
        obs = np.zeros((n, 12), self.dtype); rew = np.zeros(n, self.dtype)
        done = np.zeros(n, np.uint8); flags = np.zeros(n, np.uint8)
        term = np.full((n, 12), np.nan, self.dtype) if want_term else None
        self._f("qso_vec_step_stored_init")(C.c_int64(n), _p(rec), _p(par), _p(actions), C.c_int(kind),
                                            self.creal(dt), C.c_int(integ), C.c_int(int(auto_reset)), _p(init),
                                            _p(obs), _p(rew), _p(done), _p(flags), _p(term))
        return obs, rew, done, flags, term

    def hover_step(self, s, u_prev, a, par=PAR_NOMINAL, dt=0.02, integ=0):
        """single HoveringEnv.step -> (state', u_prev', reward, done, flags)"""
        s = self._a(s, (13,)).copy(); up = self._a(u_prev, (4,)).copy(); a = self._a(a, (4,)); par = self._a(par, (4,))
        rew = self.creal(0); done = C.c_int(0); flags = C.c_int(0)
        self._f("qso_hover_step")(_p(s), _p(up), _p(a), _p(par), self.creal(dt), C.c_int(integ),
                                  C.byref(rew), C.byref(done), C.byref(flags))
        return s, up, float(rew.value), bool(done.value), int(flags.value)

    def hover_vec_step(self, st, par, actions, init, dt=0.02, integ=0, auto_reset=True, want_term=False):
        """st [N,17] in place -> (obs[N,13], reward, done, flags, term)"""
        n = st.shape[0]
        assert st.dtype == self.dtype and st.shape == (n, 17) and st.flags.c_contiguous
        actions = self._a(actions, (n, 4)); init = self._a(init, (n, 13)); par = self._a(par, (n, 4))
        obs = np.zeros((n, 13), self.dtype); rew = np.zeros(n, self.dtype)
        done = np.zeros(n, np.uint8); flags = np.zeros(n, np.uint8)
        term = np.full((n, 13), np.nan, self.dtype) if want_term else None
        self._f("qso_hover_vec_step")(C.c_int64(n), _p(st), _p(par), _p(actions), self.creal(dt), C.c_int(integ),
                                      C.c_int(int(auto_reset)), _p(init), _p(obs), _p(rew), _p(done), _p(flags), _p(term))
        return obs, rew, done, flags, term

    def ctor_init(self, seed, gid, kind):
        """construction-time jittered initial state: kind 2 -> [26], kind 3 -> [13] (float32)"""
        out = np.zeros(26 if kind == 2 else 13, np.float32)
        self._f("qso_ctor_init")(C.c_uint64(seed), C.c_uint64(gid), C.c_int(kind), _p(out))
        return out

    def gae(self, rewards, values, dones, last_values, last_dones, gamma, lam):
        """rl_baselines/ppo2/ppo2.py:507-520 -> (advs [T,N] f32, returns [T,N] f32)"""
        rewards = np.ascontiguousarray(rewards, np.float32); values = np.ascontiguousarray(values, np.float32)
        dones = np.ascontiguousarray(dones, np.uint8); T, n = rewards.shape
        lv = np.ascontiguousarray(last_values, np.float32); ld = np.ascontiguousarray(last_dones, np.uint8)
        advs = np.zeros((T, n), np.float32); rets = np.zeros((T, n), np.float32)
        self._f("qso_gae")(C.c_int64(T), C.c_int64(n), _p(rewards), _p(values), _p(dones), _p(lv), _p(ld),
                           C.c_double(gamma), C.c_double(lam), _p(advs), _p(rets))
        return advs, rets

    def expert_action(self, sdes, sc, st, first, kp=0.35, kd=0.0, mass=0.18):
        """-> (action[4], u[4], mutated state_des[13])"""
        sdes = self._a(sdes, (13,)).copy(); sc = self._a(sc, (13,)); st = self._a(st, (13,))
        a = np.zeros(4, self.dtype); u = np.zeros(4, self.dtype)
        self._f("qso_expert_action")(_p(sdes), _p(sc), _p(st), C.c_int(int(first)), self.creal(kp), self.creal(kd),
                                     self.creal(mass), _p(a), _p(u))
        return a, u, sdes

    def sim_pid(self, T, s, sdes, par=PAR_NOMINAL, dt=0.02, integ=0, u_prev=None):
        """run_sim_PID.py:43-54 loop -> (states[T,13], u[T,4], final state, final sdes)"""
        s = self._a(s, (13,)).copy(); sdes = self._a(sdes, (13,)).copy(); par = self._a(par, (4,))
        up = np.zeros(4, self.dtype) if u_prev is None else self._a(u_prev, (4,)).copy()
        S = np.zeros((T, 13), self.dtype); U = np.zeros((T, 4), self.dtype)
        self._f("qso_sim_pid")(C.c_int64(T), _p(s), _p(up), _p(sdes), _p(par), self.creal(dt), C.c_int(integ),
                               _p(S), _p(U))
        return S, U, s, sdes

    # ---- RNG ----------------------------------------------------------------
    def philox(self, seed, subsequence, block):
        out = np.zeros(4, np.uint32)
        self._f("qso_philox4x32_10")(C.c_uint64(seed), C.c_uint64(subsequence), C.c_uint64(block), _p(out))
        return out

    def random_init(self, seed, stream, gid, ctr, rr, par_nom=PAR_NOMINAL):
        """-> (chaser[13] f32, target[13] f32, par[4] f32, u[16] f32)"""
        rr = np.asarray(rr, np.float32); pn = np.asarray(par_nom, np.float32)
        sc = np.zeros(13, np.float32); st = np.zeros(13, np.float32)
        par = np.zeros(4, np.float32); u = np.zeros(16, np.float32)
        self._f("qso_random_init")(C.c_uint64(seed), C.c_uint64(stream), C.c_uint64(gid), C.c_uint64(ctr),
                                   _p(rr), _p(pn), _p(sc), _p(st), _p(par), _p(u))
        return sc, st, par, u

    def normal4(self, seed, gid, k):
        """four standard normals of (env gid, step k): POLICY stream, Box-Muller (qso_normal4)"""
        n = np.zeros(4, self.dtype)
        self._f("qso_normal4")(C.c_uint64(seed), C.c_uint64(gid), C.c_uint64(k), _p(n))
        return n

    def random_action(self, seed, gid, k):
        a = np.zeros(4, np.float32)
        self._f("qso_random_action")(C.c_uint64(seed), C.c_uint64(gid), C.c_uint64(k), _p(a))
        return a


def rec_pack(chaser, target, u_c, u_t, qdes, last_shaping, t, dtype=np.float64):
    """assemble an env record [..., 40] from its parts"""
    chaser = np.asarray(chaser, dtype)
    rec = np.zeros(chaser.shape[:-1] + (REC_LEN,), dtype)
    rec[..., 0:13] = chaser
    rec[..., 13:26] = target
    rec[..., 26:30] = u_c
    rec[..., 30:34] = u_t
    rec[..., 34:38] = qdes
    rec[..., 38] = last_shaping
    rec[..., 39] = t
    return rec


def actor_critic_step(W, obs, noise, squash=False):
    """float64 restatement of ``model.step(obs)`` of the PPO2 MlpPolicy the reference trains (TEST INFRASTRUCTURE).
    Network: rl_baselines/common/policies.py:35-92 (mlp_extractor, net_arch [128, dict(vf=[128], pi=[128])], ReLU),
    :583-588 (vf / pi heads); ``linear`` is stable_baselines.common.tf_layers.linear (x @ w + b; package absent here).
    Distribution: rl_baselines/common/distributions.py:406-410 (neglogp), :426-430 (sample = mean + std * noise),
    :412-415 and policies.py:238-242 (the fork's tanh variant, ``squash``).
    Parity unpinned by reference outputs for value / neglogp (TensorFlow is not installed, so the reference's graph
    cannot be executed); the action mean is pinned by fixture g5 (the reference env driven by these weights).
    W: the arrays of tests/golden/policy_best_model_v0.npz.  obs [N,12], noise [N,4].
    -> (u [N,4] sampled action, value [N], neglogp [N], env_action [N,4], mean [N,4])"""
    f = lambda k: np.asarray(W[k], np.float64)                                    # noqa: E731
    obs = np.asarray(obs, np.float64)
    h = np.maximum(obs @ f("w0") + f("b0"), 0.0)
    hp = np.maximum(h @ f("w1") + f("b1"), 0.0)
    hv = np.maximum(h @ f("wv1") + f("bv1"), 0.0)
    mean = hp @ f("w2") + f("b2")
    value = (hv @ f("wv2") + f("bv2"))[:, 0]
    logstd = f("logstd").reshape(1, -1)
    std = np.exp(logstd)
    u = mean + std * np.asarray(noise, np.float64)
    neglogp = 0.5 * np.sum(np.square((u - mean) / std), axis=-1) + 0.5 * np.log(2.0 * np.pi) * u.shape[-1] \
        + np.sum(logstd, axis=-1)
    if squash:
        env_action = np.tanh(u)
        neglogp = neglogp + np.sum(np.log(1.0 - np.tanh(u) ** 2 + 1e-6), axis=1)
    else:
        env_action = np.clip(u, -1.0, 1.0)                                        # rl_baselines/ppo2/ppo2.py:483
    return u, value, neglogp, env_action, mean


def episode_stats_ref(rewards, dones, last_dones, ep_ret, ep_len):
    """Plain restatement of the episode accounting a Monitor-wrapped env feeds into Runner._run's ep_infos
    (run_docking_ppo2.py:19-35; rl_baselines/ppo2/ppo2.py:486-489): rewards [T,N]; dones [T,N] = done flag BEFORE
    step t (ppo2.py:479), last_dones [N] = flag after the last step.  ep_ret / ep_len [N] (float64 / int64) carry
    the unfinished episodes and are updated in place.  -> list of (key = t*N + env, return, length) in (step, env) order."""
    rewards = np.asarray(rewards, np.float64)
    T, n = rewards.shape
    after = np.concatenate([np.asarray(dones)[1:], np.asarray(last_dones)[None]], 0).astype(bool)
    out = []
    for t in range(T):
        ep_ret += rewards[t]
        ep_len += 1
        for i in np.nonzero(after[t])[0]:
            out.append((t * n + int(i), float(ep_ret[i]), int(ep_len[i])))
            ep_ret[i] = 0.0
            ep_len[i] = 0
    return out
