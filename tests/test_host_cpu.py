"""-m "not gpu": host logic, the C ABI's symbol table, and the world_size-2 sharding path on gloo."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "quadsim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qs_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_by_the_library():
    """the C-ABI library loads on a GPU-less host and exports every symbol include/quadsim.h declares"""
    from quadsim_amd import _lib
    _lib.build_library()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libquadsim_hip.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == declared
    assert lib.qs_version() == 131


def test_config_struct_matches_header():
    from quadsim_amd import _lib
    cfg = _lib.default_config()
    assert cfg.struct_size == ctypes.sizeof(_lib.QsConfig)
    assert (cfg.kind, cfg.num_envs, cfg.integrator, cfg.auto_reset, cfg.randomise) == (0, 1, 0, 0, 0)
    assert abs(cfg.dt - 0.02) < 1e-9 and abs(cfg.mass - 0.18) < 1e-8
    np.testing.assert_allclose(list(cfg.inertia), [0.00025, 0.000232, 0.0003738], rtol=1e-6)


def test_no_gpu_fails_loudly_and_never_falls_back():
    """without a HIP device every product entry point raises; nothing routes through the oracle"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import quadsim_amd as qa
    with pytest.raises(qa.QuadsimError):
        qa.VecDockingEnv("docking-v0", num_envs=4)
    with pytest.raises(qa.QuadsimError):
        qa.DockingEnv()
    with pytest.raises(qa.QuadsimError):
        qa.drone_step_batch(np.zeros((1, 13)), np.zeros((1, 4)), np.zeros((1, 4)))
    # the product package never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "quadsim_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "pyoracle" not in src and "libqso" not in src, f


def test_shard_range_partitions_exactly():
    from quadsim_amd import shard_range
    for total, world in ((10, 4), (65536 * 4, 4), (1048576, 8), (7, 8), (1, 1)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(262144, 3, 4) == (196608, 262144)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_spaces_match_reference_bounds():
    """docking_env.py:85-95"""
    from quadsim_amd.spaces import docking_spaces
    obs, act = docking_spaces()
    assert obs.shape == (12,) and act.shape == (4,)
    np.testing.assert_array_equal(act.low, -np.ones(4, np.float32))
    np.testing.assert_allclose(obs.high[3:], [100, 100, 100, np.pi, np.pi / 2, np.pi, 10 * np.pi, 10 * np.pi, 10 * np.pi],
                               rtol=1e-6)
    assert np.all(np.isinf(obs.high[:3]))


def test_attribute_surface_constants():
    """what the reference scripts read off env.chaser (run_trained_docking_ppo2.py:45, run_expert_policy.py:41)"""
    from quadsim_amd.drone import Drone
    d = Drone()
    L, lam = 0.086, 1.5e-9 / 6.11e-8
    np.testing.assert_allclose(d.rotor2control, [[1, 1, 1, 1], [0, L, 0, -L], [-L, 0, L, 0], [lam, -lam, lam, -lam]])
    assert d.get_arm_length() == 0.086 and d.get_mass() == 0.18 and d.dt == 0.02
    # inverse action map of run_expert_policy.py:61 round-trips through rotor2control
    u = np.array([1.9, 0.01, -0.02, 0.001])
    mean = std = 0.18 * 9.81 / 2
    a = (np.linalg.inv(d.rotor2control) @ u - mean) / std
    np.testing.assert_allclose(d.rotor2control @ (std * a + mean), u, atol=1e-12)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_sharded_rollout_matches_single_process(tmp_path):
    """world_size 2 on gloo/CPU: each rank rolls out its env shard (oracle as the stand-in stepper, keyed by
    GLOBAL env id), gathers the slabs with quadsim_amd.distributed.gather_rollout, and rank 0 checks the
    global view equals a single-process roll-out of all envs bit for bit."""
    script = tmp_path / "worker.py"
    script.write_text('''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from oracle.pyoracle import Oracle, PAR_NOMINAL
from quadsim_amd.distributed import env_shard, gather_rollout, gather_slab, rollout_global_view, split_slab, SlabGatherPipeline

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
N, T, seed = 96, 40, 5
rr = (0.5, 0.1, 0.2, 0.1, 0.8, 1.2, 0.8, 1.2)
orc = Oracle("f32")

def roll(gid0, n):
    rec = orc.env_init(n); par = np.tile(np.array(PAR_NOMINAL, np.float32), (n, 1))
    orc.vec_reset(rec, par, randomise=2, seed=seed, step_idx=0, gid0=gid0, rr=rr)
    rec[:, 39] = 570.0                      # every env times out inside the window -> resets exercised
    acts = np.stack([[orc.random_action(seed, gid0 + i, t) for i in range(n)] for t in range(T)])
    return orc.vec_rollout(rec, par, acts, kind=1, randomise=2, seed=seed, step_idx0=0, gid0=gid0, rr=rr)

lo, n = env_shard(N)
assert n == N // world and lo == rank * n
o, r, d, f = roll(lo, n)
O, R, D = gather_rollout(torch.from_numpy(o), torch.from_numpy(r), torch.from_numpy(d))
if rank == 0:
    o1, r1, d1, f1 = roll(0, N)
    assert np.array_equal(rollout_global_view(O).numpy(), o1)
    assert np.array_equal(rollout_global_view(R).numpy(), r1)
    assert np.array_equal(rollout_global_view(D).numpy(), d1)
    assert d1.sum() >= N
    # the packed-slab path: one collective
slab = torch.from_numpy(np.concatenate([o, r[..., None], d[..., None].astype(np.float32)], axis=-1).astype(np.float32))
S = gather_slab(slab)
if rank == 0:
    so, sr, sd = split_slab(rollout_global_view(S))
    assert np.array_equal(so.numpy(), o1) and np.array_equal(sr.numpy(), r1) and np.array_equal(sd.numpy(), d1.astype(bool))
# the double-buffered pipeline: roll-out k+1 produced while roll-out k is gathered; results come out in order, one call late
calls = []
def produce(buf):
    calls.append(len(calls)); buf.copy_(slab * float(len(calls)))
pipe = SlabGatherPipeline(produce, slab.shape, depth=2)
res = []
for k in range(5):
    r = pipe.step()
    assert (r is None) == (k < 2)
    if r is not None:
        assert torch.equal(r, S * float(k - 1))     # valid until the next step() call
        res.append(r.clone())
res += [x.clone() for x in pipe.flush()]
assert len(res) == 5 and calls == list(range(5)) and pipe.flush() == []
for k, G in enumerate(res):
    assert torch.equal(G, S * float(k + 1)), k
if rank == 0:
    print("OK")
dist.destroy_process_group()
''' % ROOT)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


def test_gym_id_registration_with_stub_gym(monkeypatch):
    """gym-docking/gym_docking/__init__.py:3-17 registers docking-v0/v1/v2 and hovering-v0; with a gym importable
    the same ids resolve to the HIP-backed classes (gym itself is absent on the build image: stub namespace)"""
    import types
    calls = []
    gym = types.ModuleType("gym"); envs = types.ModuleType("gym.envs"); reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda id, entry_point, **kw: calls.append((id, entry_point))
    gym.envs = envs; envs.registration = reg
    for name, mod in (("gym", gym), ("gym.envs", envs), ("gym.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, name, mod)
    from quadsim_amd.envs import register_gym_ids
    assert register_gym_ids() is True
    ids = dict(calls)
    assert ids["docking-v0"] == "quadsim_amd.envs:DockingEnv" and ids["docking-v2"] == "quadsim_amd.envs:MovingDockingEnv"
    assert ids["docking-v1"] == "quadsim_amd.envs:ImitatingDockingEnv" and ids["hovering-v0"] == "quadsim_amd.envs:HoveringEnv"
    import quadsim_amd.envs as E
    for _, ep in calls:
        assert hasattr(E, ep.split(":")[1])


def test_vecenv_and_gym_method_surface():
    """the duck-typed protocols the reference's callers use: SB2 VecEnv (rl_baselines/ppo2/ppo2.py:472-499) and old-gym Env"""
    import quadsim_amd as qa
    for m in ("reset", "step_async", "step_wait", "step", "close", "get_attr", "set_attr", "env_method", "seed", "render"):
        assert callable(getattr(qa.VecDockingEnv, m)), m
    for cls in (qa.DockingEnv, qa.MovingDockingEnv, qa.ImitatingDockingEnv, qa.HoveringEnv):
        for m in ("reset", "step", "seed", "render", "close"):
            assert callable(getattr(cls, m)), (cls, m)
    assert qa.make.__doc__ and qa.shard_range(8, 0, 2) == (0, 4)


def test_actor_critic_step_host_logic_matches_oracle():
    """ActorCriticPolicy.step / value (the per-step API's model.step on torch) against the float64 restatement, plain
    and tanh-squashed; the C struct handed to qs_runner_rollout mirrors include/quadsim.h"""
    import torch
    import quadsim_amd as qa
    from quadsim_amd import _lib
    from oracle.pyoracle import actor_critic_step
    path = os.path.join(ROOT, "tests", "golden", "policy_best_model_v0.npz")
    with np.load(path, allow_pickle=False) as z:
        W = {k: z[k] for k in z.files}
    rng = np.random.RandomState(0)
    obs = (rng.randn(200, 12) * np.array([1, 1, 1, .5, .5, .5, .3, .3, .3, .5, .5, .5])).astype(np.float32)
    noise = rng.randn(200, 4).astype(np.float32)
    for squash in (False, True):
        pol = qa.ActorCriticPolicy.from_npz(path, device="cpu", squash=squash)
        u, v, st, nl = pol.step(torch.as_tensor(obs), noise=torch.as_tensor(noise))
        ur, vr, nlr, ar, mean = actor_critic_step(W, obs, noise, squash)
        np.testing.assert_allclose(u.numpy(), ur, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(v.numpy(), vr, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(nl.numpy(), nlr, rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(pol.env_action(u).numpy(), ar, atol=1e-5)
        assert st is None
        ud = pol.step(torch.as_tensor(obs), deterministic=True)[0]
        np.testing.assert_allclose(ud.numpy(), mean, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pol.value(torch.as_tensor(obs)).numpy(), vr, rtol=1e-5, atol=1e-5)
    s = pol.c_struct()
    assert s.struct_size == ctypes.sizeof(_lib.QsActorCritic) == 104 and s.squash == 1
    np.testing.assert_allclose(list(s.logstd), W["logstd"], rtol=0, atol=0)


def test_episode_stats_reference_known_answer():
    """oracle.pyoracle.episode_stats_ref (the checker of qs_episode_stats) on a hand-worked case: 2 envs, 4 steps,
    env 0 finishes after steps 1 and 3, env 1 never; the carried episode continues in the next roll-out"""
    from oracle.pyoracle import episode_stats_ref
    rew = np.array([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0], [4.0, 40.0]])
    dones = np.array([[0, 0], [0, 0], [1, 0], [0, 0]], np.uint8)      # flags BEFORE each step: env 0 was done after step 1
    last = np.array([1, 0], np.uint8)                                   # ... and after step 3
    ep_ret, ep_len = np.zeros(2), np.zeros(2, np.int64)
    got = episode_stats_ref(rew, dones, last, ep_ret, ep_len)
    assert got == [(1 * 2 + 0, 3.0, 2), (3 * 2 + 0, 7.0, 2)]
    np.testing.assert_array_equal(ep_ret, [0.0, 100.0]); np.testing.assert_array_equal(ep_len, [0, 4])
    got = episode_stats_ref(rew[:1], dones[:1], np.array([0, 1], np.uint8), ep_ret, ep_len)
    assert got == [(1, 110.0, 5)]


def test_bench_self_launches_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the driver's form) must spawn its own ranks instead of
    asking for torch.distributed.run.  On this GPU-less host each child stops at the 'needs an MI355X' check: the
    parent has to relay that failure (non-zero exit), and must not have touched the GPU or raised the old SystemExit."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline", "--spawn-timeout", "240"], env=env, capture_output=True,
                       text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tier")
    assert "torch.distributed.run" not in p.stderr
    # (the parent stops the other rank as soon as one has failed, so one or both children get to say it)
    assert 1 <= p.stderr.count("needs an MI355X") <= 2, p.stderr[-2000:]
    assert p.returncode != 0


def test_packed_weight_images_match_the_library_layout():
    """host-side packers of the split-bf16 weight images produce exactly the byte counts the kernels copy into LDS, and
    hi + lo reproduces every weight to 2^-16 relative"""
    import quadsim_amd as qa
    from quadsim_amd import _lib
    from quadsim_amd.policy import _bf16_to_f32, pack_fast_weights
    from quadsim_amd.runner import pack_fast_actor_critic
    lib = _lib.load()
    path = os.path.join(ROOT, "tests", "golden", "policy_best_model_v0.npz")
    ac = qa.ActorCriticPolicy.from_npz(path, device="cpu")
    blob = pack_fast_actor_critic(ac)
    assert blob.size == lib.qs_runner_rollout_fast_blob_bytes()
    assert pack_fast_weights(qa.MlpPolicy.from_npz(path, device="cpu")).size == lib.qs_policy_rollout_fast_blob_bytes()
    # A2 of the policy branch: fragment [nt][p][lane][j] holds W[16 nt + c][16 (2p + (j >> 2)) + 4 g + (j & 3)]
    hi = _bf16_to_f32(blob[:32768].view(np.uint16)).reshape(8, 4, 64, 8)
    lo = _bf16_to_f32(blob[32768:65536].view(np.uint16)).reshape(8, 4, 64, 8)
    wt = ac.w1.numpy().T
    for nt, p, lane, j in [(0, 0, 0, 0), (3, 2, 37, 5), (7, 3, 63, 7)]:
        c, g = lane & 15, lane >> 4
        w = wt[16 * nt + c, 16 * (2 * p + (j >> 2)) + 4 * g + (j & 3)]
        assert abs((hi[nt, p, lane, j] + lo[nt, p, lane, j]) - w) <= abs(w) * 2.0 ** -16 + 1e-12


def test_bench_maps_gpu_count_to_baseline_configs(monkeypatch):
    """`bench.py --gpus N` must run BASELINE's config for N (c3, c3 x2, c4, c5) and say so truthfully;
    explicit flags override and drop the BASELINE label"""
    import importlib
    bench = importlib.import_module("bench")
    want = {1: ("docking-v0", 65536, 1, "config 3"), 2: ("docking-v0", 65536, 1, "config 3 per GPU (x2)"),
            4: ("docking-v2", 65536, 1, "config 4"), 8: ("docking-v2", 131072, 2, "config 5")}
    for g, (env, n, rnd, tag) in want.items():
        monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(g)])
        a = bench.parse()
        assert (a.env, a.envs_per_gpu, a.randomise, a.config_tag) == (env, n, rnd, tag)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--env", "docking-v0"])
    a = bench.parse()
    assert a.env == "docking-v0" and a.envs_per_gpu == 131072 and a.config_tag is None
