"""Pins the CPU oracle (oracle/quadsim_oracle.c, f64 build) against the golden
vectors captured from the real reference by oracle/gen_goldens.py.

Tolerance: 1e-12 absolute/relative (both sides are IEEE double; differences are
summation order only).  The f32 build of the same source is held to the
north_star's 1e-5 to show the formulas survive binary32."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.pyoracle import PAR_NOMINAL, REC_LEN, Oracle

TOL = dict(rtol=1e-12, atol=1e-12)


def test_g2_transforms(oracle64):
    g = load_golden("g2_transforms")
    for i in range(len(g["quat"])):
        np.testing.assert_allclose(oracle64.quat2euler(g["quat"][i]), g["quat2euler"][i], **TOL)
        np.testing.assert_allclose(oracle64.quat2rot(g["quat"][i]), g["quat2rot"][i], **TOL)
        np.testing.assert_allclose(oracle64.euler2quat(g["euler"][i]), g["euler2quat"][i], **TOL)
        np.testing.assert_allclose(oracle64.rot2euler(g["rot"][i]), g["rot2euler"][i], **TOL)
    # the saturation branches are present in the fixture
    assert np.sum(np.abs(g["rot"][:, 1, 2]) >= 1) > 50
    assert np.sum(np.abs(g["quat2euler"][:, 0]) == np.pi / 2) > 10


def test_g1_drone_step(oracle64):
    g = load_golden("g1_drone_step")
    n = len(g["state"])
    fired = 0
    for i in range(n):
        s2, up2, over = oracle64.drone_step(g["state"][i], g["u_prev"][i], g["u"][i], dt=float(g["dt"]))
        np.testing.assert_allclose(s2, g["state_out"][i], **TOL)
        np.testing.assert_allclose(up2, g["u_prev_out"][i], **TOL)
        assert over == int(g["limited"][i])
        fired += over
    assert fired > 500 and fired < n - 500


def test_g3_controller(oracle64):
    g = load_golden("g3_controller")
    for i in range(len(g["state_des"])):
        u, sd = oracle64.ctrl_pid(g["state_des"][i], g["state_now"][i], float(g["mass"]))
        np.testing.assert_allclose(u, g["u_pid"][i], **TOL)
        np.testing.assert_allclose(sd, g["state_des_after_pid"][i], **TOL)
        u, sd = oracle64.ctrl_vel(g["state_des"][i], g["state_now"][i], g["state_last"][i], float(g["mass"]))
        np.testing.assert_allclose(u, g["u_vel"][i], **TOL)
        np.testing.assert_allclose(sd, g["state_des_after_vel"][i], **TOL)


def _check_single_steps(orc, g, kind, par=PAR_NOMINAL, prefix="", tol=TOL, rew_atol=1e-12):
    rb, ra = g[prefix + "rec_before"], g[prefix + "rec_after"]
    for t in range(len(rb)):
        rec, obs, rew, done, flags = orc.env_step(rb[t], g[prefix + "actions"][t], par=par, kind=kind)
        np.testing.assert_allclose(rec, ra[t], **tol)
        np.testing.assert_allclose(obs, g[prefix + "obs"][t], **tol)
        assert abs(rew - g[prefix + "reward"][t]) <= rew_atol * max(1.0, abs(ra[t][38]))
        assert done == bool(g[prefix + "done"][t])
        assert (flags & 7) == int(g[prefix + "flags"][t])


@pytest.mark.parametrize("name,kind", [("g4_traj_v0", 0), ("g4_traj_v2", 1)])
def test_g4_single_steps(oracle64, name, kind):
    _check_single_steps(oracle64, load_golden(name), kind)


@pytest.mark.parametrize("name,kind", [("g4_traj_v0", 0), ("g4_traj_v2", 1)])
def test_g4_closed_loop_with_autoreset(oracle64, name, kind):
    """free-running oracle vec env (N=1, auto-reset) reproduces the whole
    1500-step reference trajectory, resets included; q_des persists over resets."""
    g = load_golden(name)
    rec = oracle64.env_init(1)
    par = np.array([PAR_NOMINAL], np.float64)
    first = oracle64.vec_reset(rec, par)
    np.testing.assert_allclose(first[0], g["first_obs"], **TOL)
    for t in range(len(g["actions"])):
        np.testing.assert_allclose(rec[0], g["rec_before"][t], rtol=1e-9, atol=1e-9)
        obs, rew, done, flags, term = oracle64.vec_step(rec, par, g["actions"][t][None], kind=kind, want_term=True)
        assert bool(done[0]) == bool(g["done"][t])
        if done[0]:
            np.testing.assert_allclose(term[0], g["obs"][t], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(obs[0], g["reset_obs"][t], rtol=1e-9, atol=1e-9)
        else:
            np.testing.assert_allclose(obs[0], g["obs"][t], rtol=1e-9, atol=1e-9)
        assert abs(rew[0] - g["reward"][t]) < 1e-9
    assert g["done"].sum() >= 10


def test_g4_known_answers():
    """spot values quoted in SURVEY.md section 8c"""
    g = load_golden("g4_traj_v0")
    np.testing.assert_allclose(g["first_obs"], [1.8] + [0.0] * 11, atol=1e-15)
    a0 = g["actions"][0].astype(np.float64)
    assert abs(g["reward"][0] - (-6.0 - 0.1 * np.linalg.norm(a0))) < 1e-12
    assert abs(g["rec_after"][0][5] - (-0.1962)) < 1e-12 and abs(g["rec_after"][0][13 + 5] - (-0.1962)) < 1e-12
    g2 = load_golden("g4_traj_v2")
    assert abs(g2["reward"][0] - (-1.8 - 0.1 * np.linalg.norm(g2["actions"][0].astype(np.float64)))) < 1e-12


def test_g5_policy_episode(oracle64):
    g = load_golden("g5_policy_episode")
    _check_single_steps(oracle64, g, 0)
    assert (g["flags"] & 1).sum() == 183 and g["flags"][-1] & 4 and g["done"][-1] and len(g["done"]) == 600


def test_g6_sim_pid(oracle64):
    g = load_golden("g6_sim_pid")
    S, U, s_fin, sdes_fin = oracle64.sim_pid(len(g["states"]), g["ini_state"], g["state_des"])
    np.testing.assert_allclose(S, g["states"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(U, g["u"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(s_fin, g["final_state"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(sdes_fin, g["final_state_des"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(s_fin[:3], [-0.2, 0.2, 0.2], atol=1e-3)


def test_g7_domain_rand(oracle64):
    g = load_golden("g7_domain_rand")
    for kind in (0, 1):
        for j in range(3):
            key = "k%d_s%d_" % (kind, j)
            _check_single_steps(oracle64, g, kind, par=tuple(g[key + "par"]), prefix=key)


@pytest.mark.parametrize("name,kind", [("g4_traj_v0", 0), ("g4_traj_v2", 1), ("g5_policy_episode", 0)])
def test_f32_build_within_north_star_tolerance(oracle32, name, kind):
    """binary32 evaluation of the same formulas from identical (f32-rounded)
    inputs stays within 1e-5 per step; done/flags never flip on these data."""
    g = load_golden(name)
    rb = g["rec_before"].astype(np.float32)
    worst = 0.0
    o64 = Oracle("f64")
    for t in range(0, len(rb), 3):
        a = g["actions"][t]
        rec64, obs64, rew64, done64, fl64 = o64.env_step(rb[t].astype(np.float64), a, kind=kind)
        rec32, obs32, rew32, done32, fl32 = oracle32.env_step(rb[t], a, kind=kind)
        np.testing.assert_allclose(rec32[:38], rec64[:38], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(obs32, obs64, rtol=1e-5, atol=2e-5)
        assert abs(rew32 - rew64) <= 1e-5 * max(1.0, abs(rec64[38]))
        assert done32 == done64 and (fl32 & 7) == (fl64 & 7)
        worst = max(worst, float(np.max(np.abs(obs32 - obs64))))
    assert worst < 2e-5


def test_philox_known_answer(oracle64):
    """Random123 known-answer test for Philox4x32-10 (counter/key all zero and
    the 'pi' vector), which rocRAND's engine implements."""
    z = oracle64.philox(0, 0, 0)
    assert [hex(int(x)) for x in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    ones = oracle64.philox(0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF)
    assert [hex(int(x)) for x in ones] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    # counter = (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), key = (0xa4093822, 0x299f31d0)
    pi = oracle64.philox(0x299f31d0a4093822, 0x0370734413198a2e, 0x85a308d3243f6a88)
    assert [hex(int(x)) for x in pi] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_random_init_ranges(oracle64):
    rr = (0.5, 0.1, 0.2, 0.1, 0.8, 1.2, 0.8, 1.2)
    U = []
    for gid in range(400):
        sc, st, par, u = oracle64.random_init(1234, 0, gid, 7, rr)
        U.append(u)
        assert np.all(np.abs(sc[0:3] - [8, -50, 5]) <= 0.5 + 1e-6) and np.all(np.abs(sc[3:6]) <= 0.1)
        assert abs(np.linalg.norm(sc[6:10]) - 1) < 1e-6 and np.all(np.abs(sc[10:13]) <= 0.1)
        np.testing.assert_array_equal(st, [10, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
        assert 0.8 * 0.18 <= par[0] <= 1.2 * 0.18 * (1 + 1e-6)
    U = np.array(U)
    assert U.min() > 0 and U.max() < 1 and abs(U.mean() - 0.5) < 0.02 and abs(U.var() - 1 / 12) < 0.01
    # lattices: three state uniforms per Philox word (11, 11 and 10 bits), the four params uniforms on 16 bits
    bits = np.array([11, 11, 10] * 4 + [16] * 4)
    scaled = U * (2.0 ** bits) - 0.5
    assert np.all(scaled == np.round(scaled)) and np.all(scaled >= 0) and np.all(scaled < 2.0 ** bits)
    # every field of every word is used and independent enough to be uniform on its own
    for j in range(16):
        assert abs(U[:, j].mean() - 0.5) < 0.06, j
    # the state uniforms come from block 2*ctr alone, the params from block 2*ctr + 1 alone
    w0 = oracle64.philox(1234, (0 << 48) | 17, 2 * 7)
    w1 = oracle64.philox(1234, (0 << 48) | 17, 2 * 7 + 1)
    _, _, _, u = oracle64.random_init(1234, 0, 17, 7, rr)
    want = []
    for w in w0:
        w = int(w)
        want += [((w & 0x7FF) + 0.5) / 2048.0, (((w >> 11) & 0x7FF) + 0.5) / 2048.0, ((w >> 22) + 0.5) / 1024.0]
    for w in w1[:2]:
        w = int(w)
        want += [((w & 0xFFFF) + 0.5) / 65536.0, ((w >> 16) + 0.5) / 65536.0]
    np.testing.assert_array_equal(u, np.array(want, np.float32))


# ---------------------------------------------------------------- section 8f-2: docking-v1, hovering-v0
def test_g8_docking_v1(oracle64):
    """docking-v1 == docking-v0 whose reset returns to the construction-time jittered chaser state"""
    g = load_golden("g8_traj_v1")
    for j in range(3):
        key = "e%d_" % j
        _check_single_steps(oracle64, g, 0, prefix=key)
        init = np.concatenate([g[key + "chaser_ini_state"], g[key + "target_ini_state"]])[None]
        rec = oracle64.env_init(1)
        rec[0, 0:13] = init[0, :13]
        np.testing.assert_allclose(oracle64.rel_obs(init[0, :13], init[0, 13:]), g[key + "first_obs"], **TOL)
        par = np.array([PAR_NOMINAL], np.float64)
        for t in range(len(g[key + "actions"])):
            np.testing.assert_allclose(rec[0], g[key + "rec_before"][t], rtol=1e-9, atol=1e-9)
            obs, rew, done, flags, term = oracle64.vec_step_stored_init(rec, par, g[key + "actions"][t][None], init,
                                                                        want_term=True)
            assert bool(done[0]) == bool(g[key + "done"][t])
            ref = g[key + "reset_obs"][t] if done[0] else g[key + "obs"][t]
            np.testing.assert_allclose(obs[0], ref, rtol=1e-9, atol=1e-9)
        assert g[key + "done"].sum() >= 10


def test_g9_hovering(oracle64):
    g = load_golden("g9_hovering")
    for j in range(3):
        key = "e%d_" % j
        sb, ub = g[key + "state_before"], g[key + "u_before"]
        for t in range(len(sb)):
            s, up, rew, done, flags = oracle64.hover_step(sb[t], ub[t], g[key + "actions"][t])
            np.testing.assert_allclose(s, g[key + "state_after"][t], **TOL)
            np.testing.assert_allclose(up, g[key + "u_after"][t], **TOL)
            assert abs(rew - g[key + "reward"][t]) < 1e-12 and done == bool(g[key + "done"][t])
        # closed loop with auto-reset to ini_state
        st = np.zeros((1, 17)); st[0, :13] = g[key + "ini_state"]
        par = np.array([PAR_NOMINAL], np.float64)
        for t in range(len(sb)):
            np.testing.assert_allclose(st[0, :13], sb[t], rtol=1e-9, atol=1e-9)
            obs, rew, done, flags, term = oracle64.hover_vec_step(st, par, g[key + "actions"][t][None],
                                                                  g[key + "ini_state"][None], want_term=True)
            assert bool(done[0]) == bool(g[key + "done"][t])
            if done[0]:
                np.testing.assert_allclose(term[0], g[key + "state_after"][t], rtol=1e-9, atol=1e-9)
                np.testing.assert_allclose(obs[0], g[key + "ini_state"], rtol=0, atol=0)
    assert g["e2_done"].sum() >= 5
    # crafted cases around the +1 bonus ball
    n_bonus = 0
    for i in range(len(g["c_state_before"])):
        s, up, rew, done, flags = oracle64.hover_step(g["c_state_before"][i], g["c_u_before"][i], g["c_actions"][i])
        np.testing.assert_allclose(s, g["c_state_after"][i], **TOL)
        assert abs(rew - g["c_reward"][i]) < 1e-12
        n_bonus += flags & 1
    assert n_bonus == int((g["c_reward"] > 1.0).sum()) and n_bonus > 30


def test_ctor_init_ranges(oracle64):
    for gid in range(200):
        d = oracle64.ctor_init(5, gid, 2)
        assert np.all(np.abs(d[0:3] - [8, -50, 5]) <= 0.3 + 1e-6) and d[6] == 1 and np.all(d[3:6] == 0)
        np.testing.assert_array_equal(d[13:], [10, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
        h = oracle64.ctor_init(5, gid, 3)
        assert np.all(np.abs(h[0:3] - [0, 0, 5]) <= 1 + 1e-6) and abs(np.linalg.norm(h[6:10]) - 1) < 1e-6
        assert np.all(h[10:] == 0) and np.all(h[3:6] == 0)


def test_g10_gae(oracle64):
    """GAE(lambda) restatement vs the reference's own lines (rl_baselines/ppo2/ppo2.py:507-520)"""
    g = load_golden("g10_gae")
    for j in range(4):
        k = "c%d_" % j
        gamma, lam = g[k + "gamma_lam"]
        advs, rets = oracle64.gae(g[k + "rewards"], g[k + "values"], g[k + "dones"], g[k + "last_values"],
                                  g[k + "last_dones"], gamma, lam)
        np.testing.assert_array_equal(advs, g[k + "advs"])            # same double recurrence, same float32 rounding
        np.testing.assert_array_equal(rets, g[k + "returns"])
        T, n = advs.shape
        np.testing.assert_array_equal(g[k + "flat_returns"], rets.swapaxes(0, 1).reshape(T * n))


def test_g11_expert_episode(oracle64):
    """PID expert (run_expert_policy.py:49-69): per-step action parity and the closed-loop episode on the oracle env"""
    g = load_golden("g11_expert_episode")
    kp, kd = g["kp_kd"]
    sdes = np.array([8, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], float)
    for t in range(len(g["actions"])):
        a, u, sdes = oracle64.expert_action(sdes, g["chaser"][t], g["target"][t], t == 0, kp, kd)
        np.testing.assert_allclose(a, g["actions"][t], **TOL)
        np.testing.assert_allclose(u, g["u"][t], **TOL)
        np.testing.assert_allclose(sdes, g["state_des_after"][t], **TOL)
    # closed loop
    rec = oracle64.env_init(1)[0]
    sdes = np.array([8, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], float)
    obs = oracle64.rel_obs(rec[0:13], rec[13:26])
    ret = 0.0
    for t in range(len(g["actions"])):
        np.testing.assert_allclose(obs, g["obs"][t], rtol=1e-8, atol=1e-8)
        a, u, sdes = oracle64.expert_action(sdes, rec[0:13], rec[13:26], t == 0, kp, kd)
        rec, obs, rew, done, flags = oracle64.env_step(rec, a)
        assert abs(rew - g["rewards"][t]) < 1e-8 and done == bool(g["done"][t])
        ret += rew
    assert abs(ret - 0.8418) < 1e-3 and done


def test_g12_dock_port_state(oracle64):
    """Drone.get_dock_port_state (quadrotor.py:213-224): position / velocity of the port for the three ports in use"""
    g = load_golden("g12_dock_port")
    for i in range(g["state"].shape[0]):
        pos, vel = oracle64.dock_port(g["state"][i], g["port"][i])
        np.testing.assert_allclose(pos, g["pos"][i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(vel, g["vel"][i], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name,kind", [("g4_traj_v0", 0), ("g4_traj_v2", 1), ("g5_policy_episode", 0)])
def test_numpy_twin_against_reference_fixtures(name, kind):
    """oracle/np_oracle.py (array restatement of SURVEY.md Appendix A, independent of the C one) on the recorded
    reference steps: 1e-12 like the C oracle, so either can serve as the checker"""
    import warnings
    from oracle import np_oracle
    g = load_golden(name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rec, obs, rew, done, flags = np_oracle.env_step(g["rec_before"], g["actions"], kind=kind)
    ra = g["rec_after"]
    np.testing.assert_allclose(rec, ra, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(obs, g["obs"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(rew, g["reward"], rtol=0, atol=1e-11)
    assert np.array_equal(done, g["done"].astype(bool))
    assert np.array_equal(flags & 7, g["flags"])


def test_numpy_twin_against_c_oracle_with_params(oracle64):
    """the two restatements against each other on the domain-randomised fixture (per-env mass / inertia)"""
    import warnings
    from oracle import np_oracle
    g = load_golden("g7_domain_rand")
    for kind in (0, 1):
        for j in range(3):
            key = "k%d_s%d_" % (kind, j)
            rb, par = g[key + "rec_before"], g[key + "par"]
            n = len(rb)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                rec, obs, rew, done, flags = np_oracle.env_step(rb, g[key + "actions"], par=np.tile(par, (n, 1)), kind=kind)
            np.testing.assert_allclose(rec, g[key + "rec_after"], rtol=1e-12, atol=1e-12)
            for i in range(0, n, 37):
                r_c, o_c, rew_c, d_c, f_c = oracle64.env_step(rb[i], g[key + "actions"][i], par=par, kind=kind)
                np.testing.assert_allclose(rec[i], r_c, rtol=1e-12, atol=1e-12)
                np.testing.assert_allclose(obs[i], o_c, rtol=1e-11, atol=1e-11)
                assert bool(done[i]) == d_c and int(flags[i]) == f_c


@pytest.mark.parametrize("squash", [False, True])
def test_actor_critic_restatement_against_independent_implementations(squash):
    """oracle.pyoracle.actor_critic_step (the float64 restatement the GPU Runner kernels are held to; value / neglogp are
    'parity unpinned' by reference outputs because TensorFlow is absent) against implementations it shares no code with: the
    three-layer heads through torch.nn.functional.linear in float64, the diagonal-Gaussian neglogp through
    scipy.stats.norm.logpdf, the tanh correction through log1p of the squared tanh; and its action mean against the actions
    fixture g5 recorded from the reference env driven by the same weights (clip(mean) = the deterministic policy)."""
    import os
    import torch
    from scipy.stats import norm
    from conftest import GOLDEN
    from oracle.pyoracle import actor_critic_step
    with np.load(os.path.join(GOLDEN, "policy_best_model_v0.npz"), allow_pickle=False) as z:
        W = {k: z[k] for k in z.files}
    rng = np.random.RandomState(7)
    obs = rng.uniform(-1.0, 1.0, (257, 12)) * np.array([3, 3, 3, 1, 1, 1, 0.5, 0.5, 0.5, 1, 1, 1.0])
    noise = rng.standard_normal((257, 4))
    u, value, neglogp, env_action, mean = actor_critic_step(W, obs, noise, squash=squash)
    t = lambda k: torch.as_tensor(np.asarray(W[k], np.float64))                      # noqa: E731
    F = torch.nn.functional
    x = torch.as_tensor(obs)
    h = F.relu(F.linear(x, t("w0").T, t("b0")))
    mean_t = F.linear(F.relu(F.linear(h, t("w1").T, t("b1"))), t("w2").T, t("b2")).numpy()
    value_t = F.linear(F.relu(F.linear(h, t("wv1").T, t("bv1"))), t("wv2").T, t("bv2")).numpy()[:, 0]
    np.testing.assert_allclose(mean, mean_t, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(value, value_t, rtol=1e-12, atol=1e-13)
    std = np.exp(np.asarray(W["logstd"], np.float64)).reshape(1, -1)
    np.testing.assert_allclose(u, mean_t + std * noise, rtol=1e-13, atol=1e-14)
    nl = -norm.logpdf(u, loc=mean_t, scale=std).sum(axis=1)
    if squash:
        nl = nl + np.log1p(-np.tanh(u) ** 2 + 1e-6).sum(axis=1)
        np.testing.assert_allclose(env_action, np.tanh(u), rtol=0, atol=1e-15)
    else:
        np.testing.assert_allclose(env_action, np.clip(u, -1.0, 1.0), rtol=0, atol=0)
    np.testing.assert_allclose(neglogp, nl, rtol=1e-11, atol=1e-11)
    # the reference itself: fixture g5 = its env driven by clip(mean(obs)) of these weights (obs[t] is the observation AFTER
    # step t; the episode starts from the nominal reset observation)
    if not squash:
        g5 = load_golden("g5_policy_episode")
        first = np.zeros((1, 12)); first[0, 0] = 1.8
        prev = np.concatenate([first, g5["obs"][:-1].astype(np.float64)])
        _, _, _, _, m = actor_critic_step(W, prev, np.zeros((len(prev), 4)))
        np.testing.assert_allclose(np.clip(m, -1, 1), g5["actions"], rtol=0, atol=5e-6)
