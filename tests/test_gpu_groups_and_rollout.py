"""-m gpu: env groups (several step chains of one handle in flight), terminal states of the VecEnv contract,
the fused GAE + flatten and episode-accounting kernels, and Runner.run() at the shape of tools/soak_runner.py.
All through the C ABI; references are the fixtures of the NumPy reference, the CPU oracle, or plain torch ops."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from helpers import OBS_TOL, STATE_TOL, set_env_from_rec, threshold_margin
from oracle.pyoracle import episode_stats_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import quadsim_amd
    return quadsim_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def _full_state(env):
    st = env.get_state()
    return np.concatenate([st["chaser"], st["target"], st["u_prev"], st["qdes"], st["last_shaping"][:, None], st["t"][:, None]], 1)


# ---------------------------------------------------------------- env groups
@pytest.mark.parametrize("n,groups,threads,env_id,rnd", [
    (65536, 2, True, "docking-v0", 1), (65536, 4, False, "docking-v0", 1), (1000, 3, True, "docking-v2", 2),
    (4096, 2, False, "docking-v0", 0), (200, 7, True, "docking-v2", 1), (262144, 2, True, "docking-v0", 1)])
def test_step_groups_bit_identical_to_single_launch(qa, torch, n, groups, threads, env_id, rnd):
    """G chains on G streams (qs_step_groups) == one qs_step launch, bit for bit: outputs of every step, terminal
    rows, the final state and the step counter -- any grouping, launcher threads or not, ragged tail tiles included"""
    kw = dict(num_envs=n, randomise=rnd, seed=11, init_range=qa.C3_INIT_RANGE, mass_scale=(0.8, 1.2),
              inertia_scale=(0.8, 1.2), env_id_offset=12345, copy=False)
    a, b = qa.VecDockingEnv(env_id, **kw), qa.VecDockingEnv(env_id, **kw)
    a.reset(); b.reset()
    t0 = np.zeros(n, np.float32); t0[::5] = 596.0          # a fifth of the envs time out inside the window
    a.set_state(t=t0); b.set_state(t=t0)
    got = b.set_groups(groups, threads=threads)
    assert got == min(groups, (n + 63) // 64)
    lo_hi = [b.group_range(g) for g in range(b.num_groups)]
    assert lo_hi[0][0] == 0 and lo_hi[-1][1] == n and all(x[1] == y[0] for x, y in zip(lo_hi, lo_hi[1:]))
    T = 6
    acts = a.random_actions(T, step0=0)
    n_done = 0
    for k in range(T):
        oa, ra, da, ia = a.step(acts[k])
        ob, rb, db = b.step_groups(acts[k])
        b.groups_join()
        torch.cuda.synchronize()
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
        assert torch.equal(a._flags, b._flags)
        d = da.cpu().numpy()
        if d.any():
            assert torch.equal(a._term[da], b._term[db]) and torch.equal(a._tstate[da], b._tstate[db])
        n_done += int(d.sum())
    assert n_done >= n // 6
    assert a.step_counter == b.step_counter == T
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    b.set_groups(1)                                          # back to one launch per step: still the same chain
    oa, _, _, _ = a.step(acts[0]); ob, _, _, _ = b.step(acts[0])
    assert torch.equal(oa, ob)
    a.close(); b.close()


def test_step_group_with_group_local_tensors_and_streams(qa, torch):
    """EnvPool-style use: every group is stepped on its own stream with its own [n_g,...] tensors, the 'policy'
    (here: a torch op producing the actions) runs on the same stream; equals the single-launch env"""
    n = 3000
    kw = dict(num_envs=n, randomise=1, seed=3, init_range=qa.C3_INIT_RANGE)
    a, b = qa.VecDockingEnv("docking-v0", **kw), qa.VecDockingEnv("docking-v0", **kw)
    a.reset(); b.reset()
    b.set_groups(3, threads=True)
    acts = a.random_actions(4, step0=0)
    for k in range(4):
        oa, ra, da, _ = a.step(acts[k])
        b.groups_fork()                                       # acts[k] was produced on the main stream
        outs = []
        for g in range(b.num_groups):
            lo, hi = b.group_range(g)
            with torch.cuda.stream(b.group_stream(g)):
                ag = (acts[k, lo:hi] * 1.0).contiguous()      # the group's policy, on the group's stream
            outs.append(b.step_group(g, ag))
        b.groups_join()
        torch.cuda.synchronize()
        ob = torch.cat([o[0] for o in outs]); rb = torch.cat([o[1] for o in outs]); db = torch.cat([o[2] for o in outs])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    a.close(); b.close()


def test_groups_survive_main_stream_calls_in_between(qa, torch):
    """reset / set_state / get_state between group steps are ordered automatically (implicit join and fork)"""
    n = 2048
    a = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=5, init_range=qa.C3_INIT_RANGE, copy=False)
    b = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=5, init_range=qa.C3_INIT_RANGE, copy=False)
    b.set_groups(2, threads=True)
    acts = a.random_actions(3, step0=0)
    for env, step in ((a, lambda e, x: e.step(x)), (b, lambda e, x: e.step_groups(x))):
        env.reset()
        step(env, acts[0])
        mask = np.zeros(n, np.uint8); mask[::3] = 1
        env.reset(mask)                                       # main-stream call right behind a group step
        step(env, acts[1])
        st = env.get_state()                                  # ... and a read-back right behind another
        env.set_state(t=st["t"] + 100.0)
        step(env, acts[2])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    assert torch.equal(a._obs, b._obs)
    a.close(); b.close()


# ---------------------------------------------------------------- VecEnv contract: terminal states, fresh outputs
@pytest.mark.parametrize("name,env_id,kind", [("g4_traj_v0", "docking-v0", 0), ("g4_traj_v2", "docking-v2", 1)])
def test_info_chaser_target_on_done_steps_are_the_terminal_states(qa, name, env_id, kind):
    """docking_env.py:226-229: info['chaser'] / info['target'] of a done step are the states OF that step.  Every
    recorded reference step is replayed as one env of an auto-resetting batch: on done rows the infos must carry the
    reference's rec_after (the state before SB2's worker resets), while the env itself already holds the reset state."""
    g = load_golden(name)
    n = len(g["rec_before"])
    env = qa.VecDockingEnv(env_id, num_envs=n, auto_reset=True)
    set_env_from_rec(env, g["rec_before"])
    obs, rew, done, infos = env.step(g["actions"])
    done = done.cpu().numpy()
    rmax = 3.0 if kind == 0 else 10.0
    safe = threshold_margin(g["obs"], g["rec_after"][:, 2], g["rec_after"][:, 39], rmax) > 1e-4
    assert np.array_equal(done[safe], g["done"][safe].astype(bool))
    idx = np.nonzero(done & safe)[0]
    assert len(idx) >= 10
    now = env.get_state()
    for i in idx:
        info = infos[int(i)]
        np.testing.assert_allclose(info["chaser"], g["rec_after"][i, 0:13], **STATE_TOL)
        np.testing.assert_allclose(info["target"], g["rec_after"][i, 13:26], **STATE_TOL)
        np.testing.assert_allclose(info["terminal_observation"], g["obs"][i], **OBS_TOL)
        np.testing.assert_allclose(now["chaser"][i, :3], [8.0, -50.0, 5.0], atol=1e-6)     # the env itself was reset
        np.testing.assert_allclose(obs[i].cpu().numpy(), g["reset_obs"][i], atol=2e-6)
    j = int(np.nonzero(~done)[0][0])
    np.testing.assert_allclose(infos[j]["chaser"], g["rec_after"][j, 0:13], **STATE_TOL)   # not done: the current state
    assert "terminal_observation" not in infos[j]
    env.close()


def test_step_outputs_are_not_aliased_by_default(qa, torch):
    n = 512
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=2, init_range=qa.C3_INIT_RANGE)
    env.reset()
    t0 = np.zeros(n, np.float32); t0[::2] = 599.0
    env.set_state(t=t0)
    acts = env.random_actions(3, step0=0)
    o1, r1, d1, i1 = env.step(acts[0])
    keep = (o1.clone(), r1.clone(), d1.clone())
    o2, r2, d2, i2 = env.step(acts[1])
    assert o1.data_ptr() != o2.data_ptr()
    assert torch.equal(o1, keep[0]) and torch.equal(r1, keep[1]) and torch.equal(d1, keep[2])
    assert d1[0].item() and "terminal_observation" in i1[0]          # the first step's infos survive the second step
    assert np.allclose(i1[0]["chaser"][6], 1.0, atol=0.1)
    with pytest.raises(qa.QuadsimError):
        i1[1]                                                         # a not-done env's state is gone after the next step
    env.close()
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=2, init_range=qa.C3_INIT_RANGE, copy=False,
                           info_state=True)
    env.reset()
    o1, _, _, i1 = env.step(acts[0]); o2, _, _, _ = env.step(acts[1])
    assert o1.data_ptr() == o2.data_ptr()                             # the documented fast path re-uses its buffers
    assert i1[1]["chaser"].shape == (13,)                             # info_state=True snapshots every step
    env.close()
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=2, init_range=qa.C3_INIT_RANGE, copy=False)
    env.reset()
    _, _, _, i1 = env.step(acts[0])
    assert i1[1]["chaser"].shape == (13,)                             # read before the next step: fine
    _, _, _, i1 = env.step(acts[1]); env.step(acts[2])
    with pytest.raises(qa.QuadsimError):
        i1[0]                                                         # copy=False without snapshots: infos live until the next step
    env.close()


def test_numpy_backend_reports_terminal_states(qa):
    g = load_golden("g4_traj_v0")
    n = len(g["rec_before"])
    env = qa.VecDockingEnv("docking-v0", num_envs=n, auto_reset=True, backend="numpy")
    set_env_from_rec(env, g["rec_before"])
    obs, rew, done, infos = env.step(g["actions"].astype(np.float32))
    safe = threshold_margin(g["obs"], g["rec_after"][:, 2], g["rec_after"][:, 39], 3.0) > 1e-4
    i = int(np.nonzero(done & safe)[0][0])
    np.testing.assert_allclose(infos[i]["chaser"], g["rec_after"][i, 0:13], **STATE_TOL)
    np.testing.assert_allclose(infos[i]["target"], g["rec_after"][i, 13:26], **STATE_TOL)
    env.close()


# ---------------------------------------------------------------- GAE + flatten in one pass, episode accounting
def test_gae_flatten_reproduces_the_reference_lines(qa, torch):
    """fixture g10 = outputs of rl_baselines/ppo2/ppo2.py:507-520,531-539 executed by the generator"""
    g = load_golden("g10_gae")
    env = qa.VecDockingEnv("docking-v0", num_envs=64)
    for c in ("c0", "c1", "c2", "c3"):
        rew, val, dn = g[c + "_rewards"], g[c + "_values"], g[c + "_dones"]
        T, n = rew.shape
        gamma, lam = [float(x) for x in g[c + "_gamma_lam"]]
        nl = np.random.RandomState(1).randn(T, n).astype(np.float32)
        out = qa.gae_and_flatten(env, torch.as_tensor(rew), torch.as_tensor(val), torch.as_tensor(nl), torch.as_tensor(dn),
                                 torch.as_tensor(g[c + "_last_values"]), torch.as_tensor(g[c + "_last_dones"]), gamma, lam,
                                 want_advs=True)
        np.testing.assert_allclose(out["advs"].cpu().numpy(), g[c + "_advs"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["returns_tm"].cpu().numpy(), g[c + "_returns"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["returns"].cpu().numpy(), g[c + "_flat_returns"], rtol=1e-5, atol=1e-5)
        sf = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1)).reshape(-1)       # noqa: E731  ppo2.py:531-539
        np.testing.assert_array_equal(out["values"].cpu().numpy(), sf(val.astype(np.float32)))
        np.testing.assert_array_equal(out["rewards"].cpu().numpy(), sf(rew.astype(np.float32)))
        np.testing.assert_array_equal(out["neglogp"].cpu().numpy(), sf(nl))
        np.testing.assert_array_equal(out["masks"].cpu().numpy(), sf(dn.astype(bool)))
    env.close()


@pytest.mark.parametrize("T,n", [(600, 1000), (37, 257), (16, 64), (15, 70), (128, 20000), (3, 5)])
def test_gae_flatten_equals_gae_plus_flatten(qa, torch, T, n):
    """ragged shapes (T not a multiple of 16 or 4, N not a multiple of 256 / 32): the one-pass kernel == qs_gae followed
    by swap_and_flatten (bit for bit against qs_gae's single-pass scan, n >= 16 384; to rounding against its two-pass
    chunked scan below that), the pass-through arrays bit for bit, and the u8 flatten == torch"""
    env = qa.VecDockingEnv("docking-v0", num_envs=64)
    rs = np.random.RandomState(T * 1000 + n)
    dev = env.device
    rew = torch.as_tensor(rs.randn(T, n).astype(np.float32)).to(dev)
    val = torch.as_tensor(rs.randn(T, n).astype(np.float32)).to(dev)
    nl = torch.as_tensor(rs.randn(T, n).astype(np.float32)).to(dev)
    dn = torch.as_tensor((rs.rand(T, n) < 0.05).astype(np.uint8)).to(dev)
    lv = torch.as_tensor(rs.randn(n).astype(np.float32)).to(dev)
    ld = torch.as_tensor((rs.rand(n) < 0.05).astype(np.uint8)).to(dev)
    out = qa.gae_and_flatten(env, rew, val, nl, dn, lv, ld, 0.99, 0.95, want_advs=True)
    advs, rets = qa.compute_gae(env, rew, val, dn, lv, ld, 0.99, 0.95)
    if n >= 16384:
        assert torch.equal(out["advs"], advs) and torch.equal(out["returns_tm"], rets)
    else:
        assert torch.allclose(out["advs"], advs, rtol=1e-5, atol=1e-5) and torch.allclose(out["returns_tm"], rets, rtol=1e-5, atol=1e-5)
    ref = lambda x: x.swapaxes(0, 1).reshape(-1)                                         # noqa: E731
    assert torch.equal(out["returns"], ref(out["returns_tm"])) and torch.equal(out["values"], ref(val))
    assert torch.equal(out["neglogp"], ref(nl)) and torch.equal(out["rewards"], ref(rew))
    assert torch.equal(out["masks"], ref(dn).bool())
    assert torch.equal(qa.swap_and_flatten(env, rets), ref(rets))
    f8 = qa.swap_and_flatten(env, dn)
    assert f8.dtype == torch.uint8 and torch.equal(f8, ref(dn))
    assert torch.equal(qa.swap_and_flatten(env, dn.bool()), ref(dn).bool())
    env.close()


def test_episode_stats_kernel_across_runs(qa, torch):
    """qs_episode_stats against the plain restatement (oracle.pyoracle.episode_stats_ref): three consecutive roll-outs
    (the second without any episode end), unfinished episodes carried on the device, (step, env) order after the sort"""
    from quadsim_amd.rollout_buffer import EpisodeTracker
    n, T = 1000, 25
    env = qa.VecDockingEnv("docking-v0", num_envs=n)
    tr = EpisodeTracker(env)
    rng = np.random.RandomState(4)
    ep_ret, ep_len = np.zeros(n), np.zeros(n, np.int64)
    for it in range(3):
        rew = rng.randn(T, n).astype(np.float32)
        p = 0.0 if it == 1 else 0.1
        dones = (rng.rand(T, n) < p).astype(np.uint8)
        last = (rng.rand(n) < p).astype(np.uint8)
        want = episode_stats_ref(rew, dones, last, ep_ret, ep_len)
        tr.update(torch.as_tensor(rew), torch.as_tensor(dones), torch.as_tensor(last))
        assert tr.count == len(want)
        ret, ln, key = [x.cpu().numpy() for x in tr.results(ordered=True)]
        assert np.array_equal(key, [w[0] for w in want])
        assert np.array_equal(ln, [w[2] for w in want])
        np.testing.assert_allclose(ret, [w[1] for w in want], atol=1e-4)
        np.testing.assert_allclose(tr.ep_ret.cpu().numpy(), ep_ret, atol=1e-4)
        assert np.array_equal(tr.ep_len.cpu().numpy(), ep_len)
        r2, l2, k2 = [x.cpu().numpy() for x in tr.results(ordered=False)]        # unordered: the same multiset
        assert np.array_equal(np.sort(k2), key)
    env.close()


# ---------------------------------------------------------------- Runner.run() at the soak shape
def _runner_run_against_torch(qa, torch, n, T, rnd, precision="f32"):
    import os
    w = os.path.join(os.path.dirname(__file__), "golden", "policy_best_model_v0.npz")
    model = qa.ActorCriticPolicy.from_npz(w)
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=rnd, seed=3, init_range=qa.C3_INIT_RANGE,
                           mass_scale=(0.9, 1.1), inertia_scale=(0.9, 1.1))
    twin = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=rnd, seed=3, init_range=qa.C3_INIT_RANGE,
                            mass_scale=(0.9, 1.1), inertia_scale=(0.9, 1.1))
    r = qa.Runner(env=env, model=model, n_steps=T, gamma=0.99, lam=0.95, collect_ep_infos=False, precision=precision)
    twin.reset()
    tot_len, tot_ret, tot_rew = 0, 0.0, 0.0
    for it in range(2):
        d_in = r.dones.clone()
        ro = qa.fused_runner_rollout(twin, model, T, dones_in=d_in, precision=precision)    # the same roll-out, raw [T,N,.]
        obs, returns, masks, actions, values, neglogp, states, ep_infos, rewards = r.run()
        torch.cuda.synchronize()
        f = lambda x: x.swapaxes(0, 1).reshape((n * T,) + tuple(x.shape[2:]))              # noqa: E731  ppo2.py:531-539
        assert torch.equal(obs, f(ro["obs"])) and torch.equal(actions, f(ro["actions"]))
        assert torch.equal(values, f(ro["values"])) and torch.equal(neglogp, f(ro["neglogp"]))
        assert torch.equal(rewards, f(ro["rewards"])) and torch.equal(masks, f(ro["dones"]).bool())
        advs, rets = qa.compute_gae(twin, ro["rewards"], ro["values"], ro["dones"], ro["last_values"], ro["last_dones"], 0.99, 0.95)
        if n >= 16384:
            assert torch.equal(returns, f(rets))          # qs_gae's single-pass scan: the same fma chain
        else:
            assert torch.allclose(returns, f(rets), rtol=1e-5, atol=1e-5)   # its two-pass chunked scan associates differently
        for x in (obs, returns, actions, values, neglogp, rewards):
            assert bool(torch.isfinite(x).all())
        # episode accounting: the count equals the number of done flags, and every env-step / every reward belongs to
        # exactly one episode (finished ones + the carried, unfinished ones)
        after = torch.cat([ro["dones"][1:], ro["last_dones"].view(1, -1)], 0).bool()
        assert r.last_ep_count == int(after.sum())
        tot_len += int(r.last_ep_lengths.long().sum()) if r.last_ep_count else 0
        tot_ret += float(r.last_ep_returns.double().sum()) if r.last_ep_count else 0.0
        tot_rew += float(ro["rewards"].double().sum())
        assert tot_len + int(r._episodes.ep_len.long().sum()) == (it + 1) * T * n
        carried = float(r._episodes.ep_ret.double().sum())
        assert abs(tot_ret + carried - tot_rew) <= 1e-3 * max(1.0, float(ro["rewards"].abs().double().sum()) * 1e-2)
        if r.last_ep_count:
            assert int(r.last_ep_lengths.min()) >= 1 and int(r.last_ep_lengths.max()) <= 600
    env.close(); twin.close()


def test_runner_run_at_the_soak_shape(qa, torch):
    """tools/soak_runner.py's shape exactly (65 536 envs x 600 steps; T = 600 leaves a 24-row tail tile in the flatten
    kernels), every returned array compared with the torch reference x.swapaxes(0,1).reshape(...)"""
    _runner_run_against_torch(qa, torch, 65536, 600, 0)


@pytest.mark.parametrize("n,T,rnd,prec", [(1000, 600, 2, "f32"), (65, 50, 1, "bf16x3"), (4097, 33, 1, "f32")])
def test_runner_run_ragged_shapes(qa, torch, n, T, rnd, prec):
    _runner_run_against_torch(qa, torch, n, T, rnd, prec)


def test_c_abi_argument_checks_of_the_new_entry_points(qa):
    lib = qa._lib.load()
    env = qa.VecDockingEnv("docking-v0", num_envs=128)
    h = env._h
    assert lib.qs_step_group(h, 0, None, None, None, None, None, None, None) != 0       # no groups configured
    assert lib.qs_set_groups(h, 65, 0) != 0
    assert lib.qs_set_groups(h, 2, 1) == 0
    k = C.c_int32(0); assert lib.qs_group_count(h, C.byref(k)) == 0 and k.value == 2
    lo, hi = C.c_int64(0), C.c_int64(0)
    assert lib.qs_group_range(h, 1, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (64, 128)
    assert lib.qs_group_range(h, 2, C.byref(lo), C.byref(hi)) != 0
    assert lib.qs_step_group(h, 5, None, None, None, None, None, None, None) != 0
    assert lib.qs_step_groups(h, None, None, None, None, None, None, None, None) != 0
    assert lib.qs_gae_flatten(h, 0, 1, *([None] * 6), 0.99, 0.95, *([None] * 7)) != 0
    assert lib.qs_episode_stats(h, 1, 1, *([None] * 6), 0, None, None, None) != 0
    assert lib.qs_set_groups(h, 1, 0) == 0
    env.close()


# ---------------------------------------------------------------- private-queue mode (qs_set_queue_mode)
@pytest.mark.parametrize("n,env_id,rnd,integ,queues", [(65536, "docking-v0", 1, "frozen", 1), (1000, "docking-v2", 2, "frozen", 1),
                                                       (4096, "docking-v0", 0, "rk4", 2), (262144, "docking-v0", 1, "frozen", 1),
                                                       (65536, "docking-v0", 1, "frozen", 2), (1000, "docking-v2", 2, "frozen", 3),
                                                       (200, "docking-v0", 1, "frozen", 4), (131072, "docking-v2", 2, "frozen", 2)])
def test_private_queue_chain_bit_identical_to_hip_stream(qa, torch, n, env_id, rnd, integ, queues):
    """step launches as hand-written AQL packets without the end-of-kernel release (the tile's state stays in its XCD's L2)
    == ordinary HIP launches, bit for bit: every step's outputs, terminal rows, the final state, the step counter; with
    main-stream calls (masked reset, set_state) in between, through both step kernels (split / serial), ragged tiles"""
    kw = dict(num_envs=n, randomise=rnd, seed=21, init_range=qa.C3_INIT_RANGE, mass_scale=(0.8, 1.2), inertia_scale=(0.8, 1.2),
              integrator=integ, copy=False)
    a, b = qa.VecDockingEnv(env_id, **kw), qa.VecDockingEnv(env_id, **kw)
    b.set_queue_mode(True, queues, ordering="host" if n == 1000 else None)    # both orderings are covered
    assert b.queue_mode == "private" and a.queue_mode == "hip-stream" and b.queue_ordering in ("stream", "host")
    a.reset(); b.reset()
    t0 = np.zeros(n, np.float32); t0[::7] = 590.0
    a.set_state(t=t0); b.set_state(t=t0)
    acts = a.random_actions(40, step0=0)
    n_done = 0
    for k in range(40):
        oa, ra, da, _ = a.step(acts[k])
        ob, rb, db, _ = b.step(acts[k])                    # step_wait drains the queue in this mode
        torch.cuda.synchronize()
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(a._flags, b._flags), k
        if bool(da.any()):
            assert torch.equal(a._term[da], b._term[db]) and torch.equal(a._tstate[da], b._tstate[db])
        n_done += int(da.sum())
        if k == 17:
            mask = np.zeros(n, np.uint8); mask[::3] = 1
            ra_, rb_ = a.reset(mask), b.reset(mask)         # a HIP-stream call in the middle of the chain
            assert torch.equal(ra_, rb_)
    assert n_done > n // 10
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    assert a.step_counter == b.step_counter == 40
    # T steps enqueued back to back with ONE drain at the end (qs_rollout_stepwise): the way the mode is meant to be used
    o1, r1, d1, f1 = a.rollout(acts[:16], stepwise=True)
    o2, r2, d2, f2 = b.rollout(acts[:16], stepwise=True)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(f1, f2)
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    b.set_queue_mode(False)                                 # and back: the chain continues on the HIP stream
    oa, _, _, _ = a.step(acts[0]); ob, _, _, _ = b.step(acts[0])
    torch.cuda.synchronize()
    assert torch.equal(oa, ob)
    a.close(); b.close()


def test_private_queue_many_steps_and_raw_loop(qa, torch):
    """the bench's use: thousands of qs_step packets with no synchronisation in between (the kernarg ring wraps), then one sync"""
    import ctypes as C
    n, K = 65536, 6000
    kw = dict(num_envs=n, randomise=1, seed=5, init_range=qa.C3_INIT_RANGE, copy=False)
    a, b = qa.VecDockingEnv("docking-v0", **kw), qa.VecDockingEnv("docking-v0", **kw)
    a.reset(); b.reset()
    b.set_queue_mode(True, 2)
    pool = a.random_actions(64, step0=0)
    p = lambda t: C.c_void_p(t.data_ptr())                 # noqa: E731
    for env in (a, b):
        lib, h = env._lib, env._h
        args = (p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term))
        torch.cuda.synchronize()
        for k in range(K):
            assert lib.qs_step(h, p(pool[k % 64]), *args) == 0
        env.sync()
    torch.cuda.synchronize()
    assert torch.equal(a._obs, b._obs) and torch.equal(a._rew, b._rew) and torch.equal(a._done, b._done)
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    assert a.step_counter == b.step_counter == K
    a.close(); b.close()


def test_private_queue_placement_guard_fails_loudly(qa, torch):
    """the mode rests on every tile being stepped by the XCD that holds it; when a workgroup finds its tile owned by another
    XCD it must touch nothing, and the next synchronising call must report it (tools/hsa_xcd_affinity_exp.py shows what
    happens without the guard: 65 472 of 65 536 envs wrong)"""
    n = 4096
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=9, init_range=qa.C3_INIT_RANGE, copy=False)
    env.reset()
    env.set_queue_mode(True)
    acts = env.random_actions(2, step0=0)
    env.step(acts[0])                                      # a good step: owners recorded
    before = _full_state(env)
    k_before = env.step_counter
    lib = qa._lib.load()
    assert lib.qs_debug_chain_poison_owner(env._h) == 0
    import ctypes as C
    p = lambda t: C.c_void_p(t.data_ptr())                 # noqa: E731
    assert lib.qs_step(env._h, p(acts[1]), p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term)) == 0
    with pytest.raises(qa.QuadsimError, match="another XCD"):
        env.sync()
    np.testing.assert_array_equal(_full_state(env), before)         # the misplaced workgroups stored nothing
    assert env.step_counter == k_before
    env.step(acts[1])                                      # the handle recovers: owners are re-learnt after the HIP-side calls above
    assert env.step_counter == k_before + 1
    env.close()


def test_private_queue_guard_sees_owner_words_written_by_other_xcds(qa, torch):
    """poisoning the owner words through hipMemset (HBM-visible) never exercised cross-XCD visibility.  Here
    the owner words are the ones the PREVIOUS packet's workgroups wrote from their own XCDs, and the next packet -- no drain,
    no HIP call in between -- runs every workgroup one tile further, i.e. on another XCD than the one that holds the tile: the
    guard must see those words (agent-scope atomics), store nothing and make the next synchronising call fail"""
    n = 8192
    lib = qa._lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())                 # noqa: E731
    for queues in (1, 2):
        env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=9, init_range=qa.C3_INIT_RANGE, copy=False)
        env.reset()
        env.set_queue_mode(True, queues)
        acts = env.random_actions(3, step0=0)
        args = (p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term))
        assert lib.qs_step(env._h, p(acts[0]), *args) == 0     # owners claimed by the XCDs that stepped the tiles
        assert lib.qs_step(env._h, p(acts[1]), *args) == 0     # ... and confirmed
        torch.cuda.synchronize(); env.sync()
        before = _full_state(env)
        k_before = env.step_counter
        assert lib.qs_step(env._h, p(acts[0]), *args) == 0     # re-claim after the HIP-side calls above
        mid = None
        assert lib.qs_debug_chain_shift_once(env._h, 1) == 0
        assert lib.qs_step(env._h, p(acts[1]), *args) == 0     # every workgroup on the wrong XCD, straight behind the claim
        with pytest.raises(qa.QuadsimError, match="another XCD"):
            env.sync()
        assert env.step_counter == k_before + 1                 # only the well-placed step counted
        a = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=9, init_range=qa.C3_INIT_RANGE, copy=False)
        a.reset()
        for k in (0, 1, 0):
            a.step(acts[k])
        np.testing.assert_array_equal(_full_state(env), _full_state(a))     # the misplaced step stored nothing
        env.step(acts[1]); a.step(acts[1])                      # the handle recovers
        np.testing.assert_array_equal(_full_state(env), _full_state(a))
        del before, mid
        env.close(); a.close()


@pytest.mark.parametrize("ordering", ["stream", "host"])
def test_private_queue_guard_reports_without_a_synchronisation(qa, torch, ordering):
    """a loop of nothing but steps (the stream-ordered mode never drains) must not run on after a misplacement: the error word is
    host memory, every submission looks at it, so qs_step itself starts failing within a few calls"""
    n = 4096
    lib = qa._lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())                 # noqa: E731
    env = qa.VecDockingEnv("docking-v0", num_envs=n, randomise=1, seed=9, init_range=qa.C3_INIT_RANGE, copy=False)
    env.reset()
    env.set_queue_mode(True, 1, ordering=ordering)
    acts = env.random_actions(2, step0=0)
    args = (p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term))
    assert lib.qs_step(env._h, p(acts[0]), *args) == 0
    assert lib.qs_debug_chain_shift_once(env._h, 1) == 0
    assert lib.qs_step(env._h, p(acts[1]), *args) == 0     # every workgroup misplaced; nobody synchronises
    failed_at = None
    for k in range(400):
        if lib.qs_step(env._h, p(acts[k & 1]), *args) != 0:
            failed_at = k
            break
    assert failed_at is not None, "400 further steps were accepted after a misplaced one"
    assert b"another XCD" in lib.qs_last_error()
    with pytest.raises(qa.QuadsimError, match="another XCD"):
        env.sync()                                          # the draining call reports it too, and re-arms the handle
    torch.cuda.synchronize()
    env.step(acts[0])                                       # usable again
    env.sync()
    env.close()


@pytest.mark.parametrize("ordering", ["stream", "host"])
def test_private_queue_follows_set_params_and_init_state(qa, torch, ordering):
    """qs_set_params / qs_set_init_state AFTER qs_set_queue_mode change the step-kernel instantiation the
    HIP-stream path picks per launch; the private queue must re-resolve it instead of stepping with nominal mass / resets"""
    n = 3000
    rs = np.random.RandomState(3)
    kw = dict(num_envs=n, randomise=0, seed=4, copy=False)
    a, b = qa.VecDockingEnv("docking-v0", **kw), qa.VecDockingEnv("docking-v0", **kw)
    b.set_queue_mode(True, 2, ordering=ordering)
    a.reset(); b.reset()
    acts = a.random_actions(60, step0=0)
    for e in (a, b):
        e.step(acts[0])                                         # the chain is open and has stepped with the nominal variant
    mass = (0.18 * rs.uniform(0.7, 1.3, n)).astype(np.float32)
    inertia = (np.array([2.5e-4, 2.32e-4, 3.738e-4]) * rs.uniform(0.7, 1.3, (n, 3))).astype(np.float32)
    ci = np.tile(np.array([8, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], np.float32), (n, 1))
    ci[:, 0:3] += rs.uniform(-0.4, 0.4, (n, 3)).astype(np.float32)
    for e in (a, b):
        e.set_params(mass, inertia)
        e.set_init_state(ci)
        e.set_state(t=np.full(n, 560.0, np.float32))            # every env times out inside the window: stored-init resets
    n_done = 0
    for k in range(1, 60):
        oa, ra, da, _ = a.step(acts[k]); ob, rb, db, _ = b.step(acts[k])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), k
        n_done += int(da.sum())
    assert n_done >= n
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    a.close(); b.close()


@pytest.mark.parametrize("n,env_id,rnd,queues", [(65536, "docking-v0", 1, 1), (65536, "docking-v0", 1, 2), (5000, "docking-v2", 2, 3)])
def test_private_queue_stream_ordered_policy_loop(qa, torch, n, env_id, rnd, queues, monkeypatch):
    """`obs -> torch policy (the shipped MlpPolicy) -> env.step` for 200 steps in private-queue mode
    with the GPU-side hand-shake (hipStreamWriteValue64 -> barrier-value packet -> completion signal -> hipStreamWaitValue64)
    and NO host synchronisation, bit-identical to the same loop on the HIP stream: every step's obs / reward / done / flags,
    the terminal rows of every finished env, the final state (rl_baselines/ppo2/ppo2.py:472-499)"""
    import os
    pol = qa.MlpPolicy.from_npz(os.path.join(os.path.dirname(__file__), "golden", "policy_best_model_v0.npz"), device="cuda:0")
    kw = dict(num_envs=n, randomise=rnd, seed=17, init_range=qa.C3_INIT_RANGE, mass_scale=(0.8, 1.2), inertia_scale=(0.8, 1.2))
    a, b = qa.VecDockingEnv(env_id, **kw), qa.VecDockingEnv(env_id, **kw)
    b.set_queue_mode(True, queues)
    if b.queue_ordering != "stream":
        pytest.skip("no stream memory operations on this device")
    oa, ob = a.reset(), b.reset()
    t0 = np.zeros(n, np.float32); t0[::3] = 450.0           # a third of the envs time out inside the window
    a.set_state(t=t0); b.set_state(t=t0)
    torch.cuda.synchronize()
    # from here on the private-queue env must not synchronise the host
    def no_sync(*a_, **k_):
        raise AssertionError("host synchronisation inside the stream-ordered step loop")
    monkeypatch.setattr(b, "sync", no_sync)
    rec_a, rec_b = [], []
    noise = torch.randn((200, n, 4), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.3
    for k in range(200):
        for env, obs, rec in ((a, oa, rec_a), (b, ob, rec_b)):
            act = torch.clamp(pol.predict(obs) + noise[k], -1.0, 1.0)      # policy kernels on torch's stream
            o, r, d, info = env.step(act)
            rec.append((o, r, d, info._dev[1], info._dev[2], info._dev[3]))
            if env is a:
                oa = o
            else:
                ob = o
    monkeypatch.undo()
    torch.cuda.synchronize()
    n_done = 0
    for k, (x, y) in enumerate(zip(rec_a, rec_b)):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2]) and torch.equal(x[3], y[3]), k
        d = x[2]
        if bool(d.any()):
            assert torch.equal(x[4][d], y[4][d]) and torch.equal(x[5][d], y[5][d]), k
        n_done += int(d.sum())
    assert n_done >= n // 3
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    assert a.step_counter == b.step_counter == 200
    a.close(); b.close()


def test_threaded_group_step_is_on_its_stream_when_the_call_returns(qa, torch):
    """with launcher threads qs_step_group only POSTED the launch; a policy kernel enqueued on the group's
    stream right after the call could overtake it.  Policy-in-the-loop on the group streams, no join between the steps."""
    n = 4096
    kw = dict(num_envs=n, randomise=1, seed=3, init_range=qa.C3_INIT_RANGE)
    a, b = qa.VecDockingEnv("docking-v0", **kw), qa.VecDockingEnv("docking-v0", **kw)
    oa = a.reset(); ob = b.reset()
    b.set_groups(2, threads=True)
    b.groups_fork()
    obs_g = []
    for g in range(b.num_groups):
        lo, hi = b.group_range(g)
        obs_g.append(ob[lo:hi].clone())
    torch.cuda.synchronize()
    pol = lambda o: torch.tanh(o[:, 3:7] * 0.7).contiguous()      # noqa: E731
    for k in range(50):
        oa, _, _, _ = a.step(pol(oa))
        for g in range(b.num_groups):
            with torch.cuda.stream(b.group_stream(g)):
                act = pol(obs_g[g])                               # reads what the previous step_group wrote, same stream
            obs_g[g] = b.step_group(g, act)[0]
    b.groups_join()
    torch.cuda.synchronize()
    assert torch.equal(oa, torch.cat(obs_g))
    np.testing.assert_array_equal(_full_state(a), _full_state(b))
    a.close(); b.close()


@pytest.mark.parametrize("n,env_id,rnd,integ", [(65536, "docking-v0", 1, "frozen"), (64 * 37 + 11, "docking-v2", 2, "frozen"),
                                                 (4096, "docking-v0", 2, "rk4"), (131072, "docking-v2", 2, "frozen")])
def test_reset_preparation_wave_is_bit_identical(qa, torch, n, env_id, rnd, integ):
    """k_env_split<.., PREP = 2> (round 3: a third wave per workgroup prepares what a rocRAND reset of the step would install, the
    chaser wave copies it) against PREP = 0 (the chaser wave expands the Philox words inside its reset branch): every output of
    every step, terminal rows, final state and per-episode params, single steps and a fused roll-out, HIP stream and private queues"""
    lib = qa._lib.load()
    lib.qs_debug_set_reset_prep.argtypes = [C.c_int]
    kw = dict(num_envs=n, randomise=rnd, seed=31, init_range=qa.C3_INIT_RANGE, mass_scale=(0.8, 1.2), inertia_scale=(0.8, 1.2),
              integrator=integ, copy=False)
    res = []
    try:
        for prep in (0, 2):
            lib.qs_debug_set_reset_prep(prep)
            env = qa.VecDockingEnv(env_id, **kw)
            env.reset()
            t0 = np.zeros(n, np.float32); t0[::4] = 585.0
            env.set_state(t=t0)
            acts = env.random_actions(48, step0=0)
            rec, n_done = [], 0
            for k in range(24):
                o, r, d, _ = env.step(acts[k])
                rec.append((o.clone(), r.clone(), d.clone(), env._flags.clone(), env._term.clone(), env._tstate.clone()))
                n_done += int(d.sum())
            env.set_queue_mode(True, 2)
            for k in range(24, 32):
                o, r, d, _ = env.step(acts[k])
                rec.append((o.clone(), r.clone(), d.clone(), env._flags.clone(), env._term.clone(), env._tstate.clone()))
            env.set_queue_mode(False)
            O, R, D, F = env.rollout(acts[32:48])
            st = _full_state(env)
            par = env.get_params() if rnd == 2 else None
            res.append((rec, (O.clone(), R.clone(), D.clone(), F.clone()), st, par, env.step_counter, n_done))
            env.close()
    finally:
        lib.qs_debug_set_reset_prep(-1)
    a, b = res
    assert a[5] >= n // 5 and a[4] == b[4] == 48
    for k, (x, y) in enumerate(zip(a[0], b[0])):
        d = x[2].bool()
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2]) and torch.equal(x[3], y[3]), k
        if bool(d.any()):
            assert torch.equal(x[4][d], y[4][d]) and torch.equal(x[5][d], y[5][d]), k
    assert all(torch.equal(u, v) for u, v in zip(a[1], b[1]))
    np.testing.assert_array_equal(a[2], b[2])
    if rnd == 2:
        np.testing.assert_array_equal(a[3][0], b[3][0]); np.testing.assert_array_equal(a[3][1], b[3][1])


def test_runner_role_split_kernel_is_bit_identical_to_one_wave_per_tile(qa):
    """k_runner_split (matrix waves + env waves, the default) against k_runner_rollout (one wave per tile) on the same envs:
    every output array and the final env state bit for bit -- exact-f32 and split-bf16 heads, plain and squashed policy,
    in-kernel normals and caller noise, rocRAND resets with and without per-env parameters, ragged env counts, carried-over
    done flags, both roll-out layouts"""
    import torch
    from test_gpu_parity import _ac_policy
    lib = qa._lib.load()
    lib.qs_debug_set_runner_serial.argtypes = [C.c_int]
    cases = [dict(n=3000, T=20, prec="f32", squash=False, rand=1, noise=False, env_major=False),
             dict(n=3000, T=20, prec="bf16x3", squash=False, rand=1, noise=True, env_major=False),
             dict(n=64 * 7 + 5, T=33, prec="f32", squash=True, rand=2, noise=False, env_major=True),
             dict(n=64 * 7 + 5, T=33, prec="bf16x3", squash=True, rand=2, noise=False, env_major=True),
             dict(n=1, T=9, prec="f32", squash=False, rand=0, noise=True, env_major=False),
             dict(n=8192, T=12, prec="bf16x3", squash=False, rand=0, noise=False, env_major=False)]
    try:
        for cs in cases:
            pol, _ = _ac_policy(qa, cs["squash"])
            res = []
            for serial in (1, 0):
                lib.qs_debug_set_runner_serial(serial)
                env = qa.VecDockingEnv("docking-v0", num_envs=cs["n"], randomise=cs["rand"], seed=11, init_range=qa.C3_INIT_RANGE)
                env.reset()
                env.set_state(t=np.full(cs["n"], 592.0, np.float32))            # every env times out (and resets) inside the roll-out
                g = torch.Generator().manual_seed(4)
                noise = torch.randn((cs["T"], cs["n"], 4), generator=g) if cs["noise"] else None
                dones_in = (torch.rand(cs["n"], generator=g) < 0.3)
                out = qa.fused_runner_rollout(env, pol, cs["T"], noise=noise, dones_in=dones_in, want_flags=True,
                                              precision=cs["prec"], env_major=cs["env_major"])
                rec = {k: v.cpu().numpy() for k, v in out.items()}
                rec.update({"state_" + k: v for k, v in env.get_state().items()})
                if cs["rand"] == 2:
                    m, inertia = env.get_params()
                    rec["mass"], rec["inertia"] = np.asarray(m), np.asarray(inertia)
                rec["counter"] = np.int64(env.step_counter)
                res.append(rec)
                env.close()
            a, b = res
            assert a["dones"].any() and a["counter"] == b["counter"]
            for k in a:
                assert np.array_equal(a[k], b[k]), (cs, k, np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max())
    finally:
        lib.qs_debug_set_runner_serial(0)
