"""PPO2 data collection on the device: the in-tree trainer's ``Runner`` (rl_baselines/ppo2/ppo2.py:438-527) over a
``VecDockingEnv``, with the whole ``for _ in range(n_steps)`` loop -- policy and value networks, Gaussian sampling,
neglogp, action clipping, ``env.step``, roll-out storage -- in ONE kernel launch (``qs_runner_rollout``), followed by
the GAE kernel and ``swap_and_flatten``.  Same class / method names and return tuple as the reference.

``ActorCriticPolicy`` is the MlpPolicy the reference trains and ships (``trained_model/best_model_v0.zip``:
shared_fc0 12->128, pi_fc0 / vf_fc0 128->128, pi 128->4, vf 128->1, ReLU, state-independent logstd;
rl_baselines/common/policies.py:35-92,:583-603).  Its ``step`` / ``value`` are plain torch (library GEMMs) and serve
the per-step API; the fused kernel evaluates the same network on the matrix cores in exact float32.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .rollout_buffer import EpisodeTracker, compute_gae, gae_and_flatten, swap_and_flatten


class ActorCriticPolicy:
    def __init__(self, weights, device="cuda", squash=False):
        import torch
        self.torch = torch
        g = lambda k: torch.as_tensor(np.ascontiguousarray(weights[k], np.float32)).to(device)  # noqa: E731
        self.w0, self.b0 = g("w0"), g("b0")            # shared_fc0
        self.w1, self.b1 = g("w1"), g("b1")            # pi_fc0
        self.w2, self.b2 = g("w2"), g("b2")            # pi
        self.wv1, self.bv1 = g("wv1"), g("bv1")        # vf_fc0
        self.wv2, self.bv2 = g("wv2"), g("bv2")        # vf
        self.logstd_host = np.asarray(weights["logstd"], np.float32).reshape(4).copy()
        self.logstd = torch.as_tensor(self.logstd_host).to(device)
        self.squash = bool(squash)
        self.initial_state = None
        self._wt = None

    @classmethod
    def from_npz(cls, path, device="cuda", squash=False):
        with np.load(path, allow_pickle=False) as z:
            return cls({k: z[k] for k in z.files}, device, squash)

    # -- model.step / model.value (policies.py:592-609) on torch ----------------------------------------------------
    def _heads(self, obs):
        t = self.torch
        h = t.relu(t.addmm(self.b0, obs, self.w0))
        mean = t.addmm(self.b2, t.relu(t.addmm(self.b1, h, self.w1)), self.w2)
        value = t.addmm(self.bv2, t.relu(t.addmm(self.bv1, h, self.wv1)), self.wv2)[:, 0]
        return mean, value

    def step(self, obs, state=None, mask=None, deterministic=False, noise=None):
        """-> (actions [N,4] (un-clipped sample), values [N], states None, neglogp [N]); `noise` [N,4] replaces the
        generator's normals (tests)"""
        t = self.torch
        mean, value = self._heads(obs)
        std = t.exp(self.logstd)
        if deterministic:
            u = mean
        else:
            eps = t.randn_like(mean) if noise is None else noise
            u = mean + std * eps
        nl = 0.5 * (((u - mean) / std) ** 2).sum(-1) + 0.5 * math.log(2.0 * math.pi) * u.shape[-1] + self.logstd.sum()
        if self.squash:
            # distributions.py:414 with 1 - tanh(u)^2 evaluated as sech(u)^2: no float32 cancellation once tanh saturates
            nl = nl + t.log(t.cosh(u.clamp(-30.0, 30.0)) ** -2 + 1e-6).sum(-1)
        return u, value, self.initial_state, nl

    def env_action(self, u):
        """what the env receives: clip to the action box (ppo2.py:483), or tanh(u) in the fork's squashed variant"""
        return self.torch.tanh(u) if self.squash else self.torch.clamp(u, -1.0, 1.0)

    def value(self, obs, state=None, mask=None):
        return self._heads(obs)[1]

    # -- device image for the fused kernel --------------------------------------------------------------------------
    def c_struct(self):
        if self._wt is None:
            tr = lambda w: w.t().contiguous()                                                  # noqa: E731
            self._wt = [tr(self.w0), self.b0.contiguous(), tr(self.w1), self.b1.contiguous(), tr(self.w2),
                        self.b2.contiguous(), tr(self.wv1), self.bv1.contiguous(), tr(self.wv2), self.bv2.contiguous()]
        s = _lib.QsActorCritic()
        s.struct_size = C.sizeof(_lib.QsActorCritic)
        s.squash = 1 if self.squash else 0
        for name, w in zip(("wt1", "b1", "wt2", "b2", "wt3", "b3", "wtv2", "bv2", "wtv3", "bv3"), self._wt):
            setattr(s, name, w.data_ptr())
        for i in range(4):
            s.logstd[i] = float(self.logstd_host[i])
        return s


def pack_fast_actor_critic(policy):
    """Weight image of qs_runner_rollout_fast (csrc/policy_rollout.hpp, 'Fast actor-critic heads'): the two 128x128
    layers and the output rows as ready-made split-bf16 A fragments in the k-order of the accumulator-as-B-operand
    chaining, the first layer and the biases in float32."""
    from .policy import _bf16_bits, _bf16_to_f32
    g = lambda t: t.detach().cpu().numpy().astype(np.float32)                          # noqa: E731
    w1t = g(policy.w0).T.copy()                       # [128][12]
    lane = np.arange(64); gg, c = lane >> 4, lane & 15
    j = np.arange(8)
    hid = lambda p: 16 * (2 * p + (j[None, :] >> 2)) + 4 * gg[:, None] + (j[None, :] & 3)      # noqa: E731  [lane][j]

    def split(a):
        hi = _bf16_bits(a)
        return hi.tobytes(), _bf16_bits(a - _bf16_to_f32(hi)).tobytes()

    parts = []
    for w in (policy.w1, policy.wv1):                 # A2 of the policy branch, then of the value branch
        wt = g(w).T                                   # [out][in]
        a2 = np.zeros((8, 4, 64, 8), np.float32)
        for nt in range(8):
            for p in range(4):
                a2[nt, p] = wt[(16 * nt + c)[:, None], hid(p)]
        parts += list(split(a2))
    g4 = np.arange(4)
    hid4 = lambda q: 16 * (2 * q + (j[None, :] >> 2)) + 4 * g4[:, None] + (j[None, :] & 3)     # noqa: E731  [g][j]
    w3t = g(policy.w2).T                              # [4][128]
    a3p = np.zeros((4, 4, 4, 8), np.float32)          # [q][row][g][j]
    for q in range(4):
        for r in range(4):
            a3p[q, r] = w3t[r][hid4(q)]
    parts += list(split(a3p))
    wv3 = g(policy.wv2)[:, 0]                         # [128]
    a3v = np.zeros((4, 4, 8), np.float32)             # [q][g][j]
    for q in range(4):
        a3v[q] = wv3[hid4(q)]
    parts += list(split(a3v))
    w1 = np.zeros((128, 13), np.float32); w1[:, :12] = w1t
    b3 = np.zeros(16, np.float32); b3[:4] = g(policy.b2); b3[4] = g(policy.bv2)[0]
    parts += [w1.tobytes(), g(policy.b0).tobytes(), g(policy.b1).tobytes(), g(policy.bv1).tobytes(), b3.tobytes()]
    return np.frombuffer(b"".join(parts), np.uint8)


def fused_runner_rollout(env, policy, T, noise=None, dones_in=None, want_flags=False, precision="f32", env_major=False):
    """qs_runner_rollout: T Runner steps for all envs in one launch, from the envs' current state.
    -> dict of device tensors: obs [T,N,12] (the observations acted on), actions [T,N,4] (un-clipped samples),
    values, neglogp, rewards [T,N] f32, dones [T,N] u8 (flags BEFORE each step), flags [T,N] u8 or None,
    last_obs [N,12], last_values [N], last_dones [N] u8.
    precision "f32": exact-float32 MFMA; "bf16x3": qs_runner_rollout_fast, split-bf16 operands on the bf16 matrix
    rate, ~1e-5 error on means / values (opt-in).
    env_major: obs / actions come out as [N,T,12] / [N,T,4] (what swap_and_flatten would make of them); the [T,N]
    scalars are unaffected."""
    import torch
    n, dev = env.num_envs, env.device
    f32, u8 = torch.float32, torch.uint8
    out = {
        "obs": torch.empty((n, T, 12) if env_major else (T, n, 12), dtype=f32, device=dev),
        "actions": torch.empty((n, T, 4) if env_major else (T, n, 4), dtype=f32, device=dev),
        "values": torch.empty((T, n), dtype=f32, device=dev), "neglogp": torch.empty((T, n), dtype=f32, device=dev),
        "dones": torch.empty((T, n), dtype=u8, device=dev), "rewards": torch.empty((T, n), dtype=f32, device=dev),
        "flags": torch.empty((T, n), dtype=u8, device=dev) if want_flags else None,
        "last_obs": torch.empty((n, 12), dtype=f32, device=dev), "last_values": torch.empty((n,), dtype=f32, device=dev),
        "last_dones": torch.empty((n,), dtype=u8, device=dev),
    }
    if noise is not None:
        noise = noise.to(device=dev, dtype=f32).contiguous()
        if tuple(noise.shape) != (T, n, 4):
            raise ValueError("noise must be [T, num_envs, 4]")
    if dones_in is not None:
        dones_in = dones_in.to(device=dev).to(u8).contiguous()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    env._use_current_stream()
    env._inputs_ready()
    _lib.check(env._lib.qs_set_rollout_layout(env._h, 1 if env_major else 0), "qs_set_rollout_layout")
    tail = (p(noise), p(dones_in), p(out["obs"]), p(out["actions"]), p(out["values"]), p(out["neglogp"]), p(out["dones"]),
            p(out["rewards"]), p(out["flags"]), p(out["last_obs"]), p(out["last_values"]), p(out["last_dones"]))
    if precision == "f32":
        pol = policy.c_struct()
        _lib.check(env._lib.qs_runner_rollout(env._h, T, C.byref(pol), *tail), "qs_runner_rollout")
    elif precision == "bf16x3":
        if getattr(policy, "_ac_blob", None) is None:
            blob = pack_fast_actor_critic(policy)
            assert blob.size == env._lib.qs_runner_rollout_fast_blob_bytes()
            policy._ac_blob = torch.as_tensor(blob.copy()).to(dev)
        ls = (C.c_float * 4)(*[float(x) for x in policy.logstd_host])
        _lib.check(env._lib.qs_runner_rollout_fast(env._h, T, p(policy._ac_blob), ls, 1 if policy.squash else 0, *tail),
                   "qs_runner_rollout_fast")
    else:
        raise ValueError("precision must be 'f32' or 'bf16x3'")
    env._outputs_ready()
    return out


class Runner:
    """rl_baselines/ppo2/ppo2.py:438-527.  ``run()`` returns the reference's 9-tuple
    (obs, returns, masks, actions, values, neglogpacs, states, ep_infos, true_reward), every array flattened env-major by
    ``swap_and_flatten`` and resident on the device.  ``reset_after_run`` reproduces the fork's ``self.obs =
    self.env.reset()`` after every roll-out (:525); stable-baselines proper keeps the envs running (default).
    ``ep_infos`` lists {'r': episode return, 'l': length} of the episodes that ended inside the roll-out (what the
    Monitor wrapper of run_docking_ppo2.py:19-35 reports through ``info['episode']``); with tens of thousands of envs
    prefer ``collect_ep_infos=False`` and read the device tensors ``last_ep_returns`` / ``last_ep_lengths``."""

    def __init__(self, *, env, model, n_steps, gamma, lam, reset_after_run=False, collect_ep_infos=True,
                 track_episodes=True, precision="f32", fused=None):
        import torch
        self.torch = torch
        self.env, self.model, self.n_steps, self.gamma, self.lam = env, model, int(n_steps), float(gamma), float(lam)
        self.reset_after_run = reset_after_run
        self.precision = precision                     # "f32" | "bf16x3" (fused_runner_rollout)
        # fused: the whole n_steps loop in one launch (docking envs + the shipped MlpPolicy architecture); None = whenever
        # it applies, else the spelt-out loop (model.step on torch + env.step)
        can_fuse = isinstance(model, ActorCriticPolicy) and env.obs_dim == 12 and getattr(env, "auto_reset", True)
        self.fused = can_fuse if fused is None else bool(fused)
        self.collect_ep_infos = collect_ep_infos       # build the reference's list of {'r', 'l'} dicts on the host
        self.track_episodes = track_episodes           # keep episode returns / lengths at all (device tensors
        #                                                last_ep_returns / last_ep_lengths after each run)
        self.obs = env.reset()
        self.states = model.initial_state
        self.dones = torch.zeros((env.num_envs,), dtype=torch.uint8, device=env.device)
        self._episodes = EpisodeTracker(env) if track_episodes else None
        self.num_timesteps = 0

    @property
    def last_ep_returns(self):
        """returns of the episodes that ended in the last run(), (step, env) order (device tensor)"""
        return self._episodes.results()[0] if self._episodes is not None and self._episodes._bufs is not None else None

    @property
    def last_ep_lengths(self):
        return self._episodes.results()[1] if self._episodes is not None and self._episodes._bufs is not None else None

    @property
    def last_ep_count(self):
        return self._episodes.count if self._episodes is not None and self._episodes._bufs is not None else 0

    def _episode_infos(self, rewards, dones, last_dones):
        """returns / lengths of the episodes that ended inside this roll-out: ONE kernel (qs_episode_stats), no [T,N]
        temporaries; the reference's list of dicts is only built on request"""
        self._episodes.update(rewards, dones, last_dones)
        if not self.collect_ep_infos:
            return []
        ret, ln, _ = self._episodes.results(ordered=True)
        r = ret.cpu().numpy()
        l_ = ln.cpu().numpy()
        return [{"r": float(a), "l": int(b)} for a, b in zip(r, l_)]

    def _stepwise_rollout(self, noise=None):
        """the reference loop spelt out (ppo2.py:472-499): model.step on torch, env.step per step.  For envs / policies the
        fused kernel does not cover (hovering-v0, stored initial states, any other network): still no host round trip."""
        t = self.torch
        env, T = self.env, self.n_steps
        obs = self.obs if self.obs is not None else env.reset()
        dones = self.dones.bool()
        lo = t.as_tensor(env.action_space.low, device=env.device); hi = t.as_tensor(env.action_space.high, device=env.device)
        O, A, V, NL, D, R = [], [], [], [], [], []
        for k in range(T):
            kw = {} if noise is None else {"noise": noise[k]}
            u, v, self.states, nl = self.model.step(obs, self.states, dones, **kw)
            O.append(obs.clone()); A.append(u); V.append(v); NL.append(nl); D.append(dones.to(t.uint8))
            a_env = self.model.env_action(u) if hasattr(self.model, "env_action") else t.max(t.min(u, hi), lo)   # :483
            obs, r, dones, _ = env.step(a_env)
            obs, dones = obs.clone(), dones.clone()
            R.append(r.clone())
        return {"obs": t.stack(O), "actions": t.stack(A), "values": t.stack(V), "neglogp": t.stack(NL), "dones": t.stack(D),
                "rewards": t.stack(R), "last_obs": obs, "last_values": self.model.value(obs, self.states, dones),
                "last_dones": dones.to(t.uint8)}

    def run(self, noise=None):
        t = self.torch
        env, T = self.env, self.n_steps
        mb_states = self.states
        if self.fused:
            # obs / actions leave the kernel env-major: no transpose pass over the two wide arrays
            ro = fused_runner_rollout(env, self.model, T, noise=noise, dones_in=self.dones, precision=self.precision,
                                      env_major=True)
            flat_obs, flat_actions = ro["obs"].view(T * env.num_envs, 12), ro["actions"].view(T * env.num_envs, 4)
        else:
            ro = self._stepwise_rollout(noise)
            flat_obs, flat_actions = swap_and_flatten(env, ro["obs"]), swap_and_flatten(env, ro["actions"])
        self.num_timesteps += T * env.num_envs
        # ppo2.py:507-523 in one pass: GAE + the env-major flatten of returns / dones / values / neglogp / rewards
        gf = gae_and_flatten(env, ro["rewards"], ro["values"], ro["neglogp"], ro["dones"], ro["last_values"],
                             ro["last_dones"], self.gamma, self.lam)
        ep_infos = []
        if self.track_episodes:
            ep_infos = self._episode_infos(ro["rewards"], ro["dones"], ro["last_dones"])
        self.obs, self.dones = ro["last_obs"], ro["last_dones"]
        out = (flat_obs, gf["returns"], gf["masks"], flat_actions, gf["values"], gf["neglogp"], mb_states, ep_infos,
               gf["rewards"])
        if self.reset_after_run:
            self.obs = env.reset()                                                         # ppo2.py:525
        return out
