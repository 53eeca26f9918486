"""VecDockingEnv -- N parallel docking-v0 / docking-v2 envs resident on one MI355X.

Host-side mirror of what a trainer sees of the reference: the SB2 ``VecEnv``
protocol that ``SubprocVecEnv([make_env(...)]*10)`` provides in
run_docking_ppo2.py:65-67 and whose contract is visible in the in-tree Runner,
rl_baselines/ppo2/ppo2.py:472-499 (``obs, rewards, dones, infos =
env.step(clipped_actions)``; auto-reset on done with
``infos[i]['terminal_observation']``).  All arithmetic happens in the fused HIP
kernels behind the C ABI (include/quadsim.h); torch is only the owner of the
device buffers and the stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from .spaces import Box, docking_spaces, hovering_spaces

_KINDS = {"docking-v0": _lib.KIND_V0, "docking-v2": _lib.KIND_V2, "docking-v1": _lib.KIND_V1,
          "hovering-v0": _lib.KIND_HOVER}
_KINDS.update({"gym_docking:" + k: v for k, v in list(_KINDS.items())})
_INTEG = {"frozen": _lib.INTEG_FROZEN, "rk4": _lib.INTEG_RK4}

# chaser initial-state jitter of BASELINE config 3: the ranges commented out at docking_env.py:34-37
C3_INIT_RANGE = (0.5, 0.1, 0.2, 0.1)


def _torch():
    import torch
    return torch


def shard_range(num_envs_total, rank, world_size):
    """[start, stop) of the env ids owned by `rank` when envs are partitioned contiguously
    over `world_size` GPUs (envs are independent: no data-path collective)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(int(num_envs_total), world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class VecDockingEnv:
    """N envs on one GPU behind the VecEnv protocol.

    backend='torch': actions / obs / rewards / dones are CUDA tensors (zero copy).
    backend='numpy': host arrays in and out (what SB2 expects); one D2H copy per step.
    """

    metadata = {"render.modes": ["human"]}

    def __init__(self, env_id="docking-v0", num_envs=1, device=0, integrator="frozen", dt=0.02,
                 auto_reset=True, randomise=0, seed=0, env_id_offset=0, init_range=(0.0, 0.0, 0.0, 0.0),
                 mass_scale=(1.0, 1.0), inertia_scale=(1.0, 1.0), mass=0.18,
                 inertia=(0.00025, 0.000232, 0.0003738), backend="torch", use_torch_stream=True, copy=True,
                 info_state=False):
        if env_id not in _KINDS:
            raise ValueError("unknown env id %r (have %s)" % (env_id, sorted(_KINDS)))
        if backend not in ("torch", "numpy"):
            raise ValueError("backend must be 'torch' or 'numpy'")
        self.env_id = env_id
        self.kind = _KINDS[env_id]
        self.num_envs = int(num_envs)
        self.backend = backend
        self.device_index = int(device)
        self.obs_dim = 13 if self.kind == _lib.KIND_HOVER else 12
        self.observation_space, self.action_space = (hovering_spaces() if self.kind == _lib.KIND_HOVER
                                                     else docking_spaces())
        self._lib = _lib.load()
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.QuadsimError("VecDockingEnv needs a HIP device (torch.cuda.is_available() is False); "
                                    "there is no CPU path")
        self.device = torch.device("cuda", self.device_index)
        cfg = _lib.default_config()
        cfg.kind = self.kind
        cfg.num_envs = self.num_envs
        cfg.device = self.device_index
        cfg.integrator = _INTEG[integrator]
        cfg.dt = dt
        cfg.auto_reset = 1 if auto_reset else 0
        cfg.randomise = int(randomise)
        cfg.io_space = _lib.IO_DEVICE
        cfg.seed = int(seed)
        cfg.env_id_offset = int(env_id_offset)
        cfg.init_range = (C.c_float * 4)(*init_range)
        cfg.mass_scale = (C.c_float * 2)(*mass_scale)
        cfg.inertia_scale = (C.c_float * 2)(*inertia_scale)
        cfg.mass = mass
        cfg.inertia = (C.c_float * 3)(*inertia)
        # copy=True (default): step() hands out tensors nobody else writes to -- every step's outputs are written by the
        # kernel straight into freshly allocated tensors (no device copy), as SB2's VecEnvs hand out fresh arrays.
        # copy=False: views of the env's own buffers -- and the step's infos -- valid until the next step (saves ~6 allocator
        # calls per step; a step() call costs ~7 us on the host instead of ~17).
        self.copy = bool(copy)
        # info_state=True: infos[i]['chaser'/'target'] of envs that did NOT finish are snapshotted at every step (one
        # extra kernel); False: they are fetched when first asked for, which must happen before the next step
        self.info_state = bool(info_state)
        self._follow_torch_stream = bool(use_torch_stream)
        self._stream = None
        if use_torch_stream:
            self._stream = torch.cuda.current_stream(self.device).cuda_stream
            cfg.stream = self._stream or None
            cfg.external_stream = 1
        self.cfg = cfg
        self._h = C.c_void_p()
        _lib.check(self._lib.qs_create(C.byref(cfg), C.byref(self._h)), "qs_create")
        n = self.num_envs
        kw = dict(device=self.device)
        self._obs = torch.empty((n, self.obs_dim), dtype=torch.float32, **kw)
        self._rew = torch.empty((n,), dtype=torch.float32, **kw)
        self._done = torch.empty((n,), dtype=torch.uint8, **kw)
        self._flags = torch.empty((n,), dtype=torch.uint8, **kw)
        self._term = torch.zeros((n, self.obs_dim), dtype=torch.float32, **kw)
        self._tstate = None if self.kind == _lib.KIND_HOVER else torch.zeros((n, 26), dtype=torch.float32, **kw)
        self._actions = None
        self._nstep = 0                                 # steps issued (InfoView staleness guard)
        self._groups = None
        self._queue_private, self._queue_host_ordered = False, False
        self._pin, self._pin_i, self._act_pin = None, 0, None
        self.auto_reset = bool(auto_reset)
        # attribute surface the reference scripts poke (run_trained_docking_ppo2.py:45)
        self.action_mean = np.ones(4) * mass * 9.81 / 2.0
        self.action_std = np.ones(4) * mass * 9.81 / 2.0

    # ------------------------------------------------------------------ plumbing
    def _ptr(self, t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def _as_device(self, x, shape, dtype=None):
        torch = _torch()
        dtype = dtype or torch.float32
        if (isinstance(x, torch.Tensor) and x.dtype == dtype and x.device == self.device and x.is_contiguous()
                and tuple(x.shape) == tuple(shape)):
            return x                                   # the per-step fast path: nothing to convert
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=dtype)
        else:
            t = torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(self.device)
        t = t.contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    def _out(self, t):
        return t.cpu().numpy() if self.backend == "numpy" else t

    def _use_current_stream(self):
        """launch on whatever stream torch considers current (it changes under torch.cuda.stream(...) and during
        torch.cuda.graph capture); a no-op when it has not changed"""
        if self._follow_torch_stream:
            raw = getattr(_torch()._C, "_cuda_getCurrentRawStream", None)       # ~0.2 us; the public API costs ~2 us
            s = raw(self.device_index) if raw is not None else _torch().cuda.current_stream(self.device).cuda_stream
            if s != self._stream:
                _lib.check(self._lib.qs_set_stream(self._h, C.c_void_p(s) if s else None, 1), "qs_set_stream")
                self._stream = s

    def _outputs_ready(self):
        """an owned (non-torch) stream is not ordered with torch's: drain it before torch touches what a kernel wrote"""
        if not self._follow_torch_stream:
            self.sync()

    def _inputs_ready(self):
        """... and make sure torch has finished producing the inputs before a kernel on the owned stream reads them"""
        if not self._follow_torch_stream:
            _torch().cuda.current_stream(self.device).synchronize()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.qs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.check(self._lib.qs_sync(self._h), "qs_sync")

    # ------------------------------------------------------------------ VecEnv protocol
    def reset(self, mask=None):
        """DockingEnv.reset for all (or masked) envs -> obs [N,12]"""
        self._use_current_stream()
        m = None
        if mask is not None:
            m = self._as_device(mask, (self.num_envs,), _torch().uint8)
        self._inputs_ready()
        _lib.check(self._lib.qs_reset(self._h, self._ptr(m), self._ptr(self._obs)), "qs_reset")
        self._outputs_ready()
        return self._out(self._obs.clone() if self.backend == "torch" else self._obs)

    def step_async(self, actions):
        self._use_current_stream()
        torch = _torch()
        if self.backend == "numpy" and isinstance(actions, np.ndarray):
            # host actions: through a pinned staging tensor, asynchronously (no pageable-memory copy on the way up)
            if self._act_pin is None:
                self._act_pin = torch.empty((self.num_envs, 4), dtype=torch.float32, pin_memory=True)
                self._act_dev = torch.empty((self.num_envs, 4), dtype=torch.float32, device=self.device)
            np.copyto(self._act_pin.numpy(), actions.reshape(self.num_envs, 4), casting="same_kind")
            self._act_dev.copy_(self._act_pin, non_blocking=True)
            self._actions = self._act_dev
        else:
            self._actions = self._as_device(actions, (self.num_envs, 4))
        if self.copy and self.backend == "torch":
            # this step's outputs get tensors of their own: whatever the caller kept from earlier steps stays intact
            n, kw = self.num_envs, dict(device=self.device)
            self._obs = torch.empty((n, self.obs_dim), dtype=torch.float32, **kw)
            self._rew = torch.empty((n,), dtype=torch.float32, **kw)
            self._done = torch.empty((n,), dtype=torch.uint8, **kw)
            self._flags = torch.empty((n,), dtype=torch.uint8, **kw)
            if self.auto_reset:
                self._term = torch.empty((n, self.obs_dim), dtype=torch.float32, **kw)
                if self._tstate is not None:
                    self._tstate = torch.empty((n, 26), dtype=torch.float32, **kw)
        self._inputs_ready()
        if self._queue_host_ordered:
            _torch().cuda.current_stream(self.device).synchronize()      # the queue is not ordered behind the caller's stream
        _lib.check(self._lib.qs_step_ex(self._h, self._ptr(self._actions), self._ptr(self._obs), self._ptr(self._rew),
                                        self._ptr(self._done), self._ptr(self._flags),
                                        self._ptr(self._term) if self.auto_reset else None,
                                        self._ptr(self._tstate) if self.auto_reset else None), "qs_step")
        self._nstep += 1

    def step_wait(self):
        self._outputs_ready()
        if self._queue_host_ordered:
            self.sync()                                                    # outputs of a host-ordered private-queue step: valid after the drain
        if self.backend == "torch":
            # done is the uint8 buffer seen as bool.  The InfoView keeps THIS step's tensors (done, flags, terminal
            # observation / states): with copy=True they are never written again, so it can be read at any time
            infos = InfoView(self, done_t=self._done, flags_t=self._flags, term_t=self._term, tstate_t=self._tstate)
            return self._obs, self._rew, self._done.view(_torch().bool), infos
        # numpy backend: async copies into pinned host mirrors, ONE stream sync.  Two sets of mirrors alternate,
        # so the arrays of a step stay valid until the step after the next one (SB2's runner copies them at once).
        torch = _torch()
        srcs = [self._obs, self._rew, self._done, self._flags, self._term] + ([self._tstate] if self._tstate is not None else [])
        if self._pin is None:
            mk = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True)                 # noqa: E731
            self._pin = [[mk(t) for t in srcs] for _ in range(2)]
        self._pin_i ^= 1
        h = self._pin[self._pin_i]
        for dst, src in zip(h, srcs[:4] if not self.auto_reset else srcs):
            dst.copy_(src, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        obs, rew, done, flags = h[0].numpy(), h[1].numpy(), h[2].numpy().view(np.bool_), h[3].numpy()
        if self.copy:
            obs, rew, done = obs.copy(), rew.copy(), done.copy()
        # terminal rows are only meaningful where done: copy those out of the mirror (it is recycled two steps later)
        term = tstate = None
        if self.auto_reset and done.any():
            term = h[4].numpy().copy()
            tstate = h[5].numpy().copy() if self._tstate is not None else None
        infos = InfoView(self, done=done, flags=flags.copy(), term=term, tstate=tstate)
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def step_policy(self, policy, precision="f32"):
        """ONE launch per step of the loop `action, _ = model.predict(obs, deterministic=True); obs, r, done, info =
        env.step(action)` (run_trained_docking_ppo2.py:37-60): the MlpPolicy actor on the matrix cores (exact-f32 MFMA, or
        split-bf16 with precision="bf16x3"), fed with state2rel of the envs' CURRENT state (= the observation the previous step
        or reset returned), then the env step -- qs_policy_rollout with T = 1 on the caller's stream, outputs consumable in
        stream order.  65 536 envs: 32 us (f32) / 17 us (bf16x3) per step against 85-94 us for step(policy.predict(obs)).
        Returns (obs, reward, done, actions); flags in `env.last_flags`.  No terminal_observation (use step() where needed).
        torch backend, docking envs."""
        from .policy import pack_fast_weights
        torch = _torch()
        if self.backend != "torch":
            raise _lib.QuadsimError("step_policy: torch backend only")
        self._use_current_stream()
        n, kw = self.num_envs, dict(device=self.device)
        if self.copy:
            self._obs = torch.empty((n, self.obs_dim), dtype=torch.float32, **kw)
            self._rew = torch.empty((n,), dtype=torch.float32, **kw)
            self._done = torch.empty((n,), dtype=torch.uint8, **kw)
            self._flags = torch.empty((n,), dtype=torch.uint8, **kw)
        acts = torch.empty((n, 4), dtype=torch.float32, **kw)
        self._inputs_ready()
        io = (self._ptr(self._obs), self._ptr(self._rew), self._ptr(self._done), self._ptr(self._flags), self._ptr(acts))
        if precision == "bf16x3":
            if not hasattr(policy, "_blob"):
                blob = pack_fast_weights(policy)
                assert blob.size == self._lib.qs_policy_rollout_fast_blob_bytes()
                policy._blob = torch.as_tensor(blob.copy()).to(self.device)
            _lib.check(self._lib.qs_policy_rollout_fast(self._h, 1, self._ptr(policy._blob), *io), "qs_policy_rollout_fast")
        elif precision == "f32":
            if not hasattr(policy, "_wt"):
                policy._wt = [policy.w0.t().contiguous(), policy.b0.contiguous(), policy.w1.t().contiguous(),
                              policy.b1.contiguous(), policy.w2.t().contiguous(), policy.b2.contiguous()]
            _lib.check(self._lib.qs_policy_rollout(self._h, 1, *[self._ptr(w) for w in policy._wt], *io), "qs_policy_rollout")
        else:
            raise ValueError("precision must be 'f32' or 'bf16x3'")
        self._nstep += 1
        self._outputs_ready()
        return self._obs, self._rew, self._done.view(torch.bool), acts

    @property
    def last_flags(self):
        """per-env flag bits of the latest step (QS_FLAG_*: 1 docked, 2 over limit, 4 over time, 8 / 16 attitude limiter fired)"""
        return self._flags

    # ------------------------------------------------------------------ private-queue mode
    def set_queue_mode(self, private=True, queues=1, ordering=None):
        """qs_set_queue_mode: step launches go to an AQL queue owned by the handle, WITHOUT the end-of-kernel cache write-back
        HIP attaches to every launch (1.6 of 6.5 us per step at 65 536 envs); results are bit-identical.  The handle's own
        calls stay ordered (anything but a step drains the queue).  ordering: "stream" (default where the device has stream
        memory operations): a step is ordered against the env's stream -- torch's current stream -- by a GPU-side hand-shake
        (hipStreamWriteValue64 -> barrier-value packet -> completion signal -> hipStreamWaitValue64), its outputs are stored
        write-through, and step() / rollout(stepwise=True) involve NO host synchronisation: `obs -> policy -> step` loops run
        as in the default mode.  "host" (round 2's contract): no hand-shake; tensors handed to step() must be complete when
        it is called and its outputs are valid after sync() -- step_async / step_wait then synchronise the host per step.
        queues (1..4): split the tiles over that many private queues; their chains overlap each other's kernel boundary."""
        self._use_current_stream()
        _lib.check(self._lib.qs_set_queue_mode(self._h, int(queues) if private else 0), "qs_set_queue_mode")
        self._queue_private = bool(private)
        self._queue_host_ordered = False
        if private:
            if ordering is not None:
                if ordering not in ("stream", "host"):
                    raise ValueError("ordering must be 'stream' or 'host'")
                _lib.check(self._lib.qs_set_queue_ordering(self._h, _lib.ORDER_STREAM if ordering == "stream" else _lib.ORDER_HOST),
                           "qs_set_queue_ordering")
            o = C.c_int32(0)
            _lib.check(self._lib.qs_get_queue_ordering(self._h, C.byref(o)), "qs_get_queue_ordering")
            self._queue_host_ordered = o.value == _lib.ORDER_HOST

    @property
    def queue_mode(self):
        return "private" if self._queue_private else "hip-stream"

    @property
    def queue_ordering(self):
        """'stream' / 'host' in private-queue mode, else None (HIP-stream launches are stream-ordered by nature)"""
        return None if not self._queue_private else ("host" if self._queue_host_ordered else "stream")

    # ------------------------------------------------------------------ env groups (EnvPool-style send / recv)
    def set_groups(self, groups, threads=True):
        """Partition the envs into `groups` contiguous groups, each stepped on its own stream (qs_set_groups): the step
        of one group overlaps the kernel boundary -- or the policy -- of the others.  groups <= 1 removes the grouping.
        threads: one native launcher thread per group (the calling thread only posts launch records).
        Results are bit-identical to step() for any grouping."""
        self._use_current_stream()
        _lib.check(self._lib.qs_set_groups(self._h, int(groups), 1 if threads else 0), "qs_set_groups")
        self._groups = None
        k = C.c_int32(0)
        _lib.check(self._lib.qs_group_count(self._h, C.byref(k)), "qs_group_count")
        if groups > 1:
            torch = _torch()
            self._groups = []
            for g in range(k.value):
                lo, hi, sp = C.c_int64(0), C.c_int64(0), C.c_void_p()
                _lib.check(self._lib.qs_group_range(self._h, g, C.byref(lo), C.byref(hi)), "qs_group_range")
                _lib.check(self._lib.qs_group_stream(self._h, g, C.byref(sp)), "qs_group_stream")
                self._groups.append({"range": (lo.value, hi.value), "stream_ptr": sp.value,
                                     "stream": torch.cuda.ExternalStream(sp.value, device=self.device)})
        return k.value

    @property
    def num_groups(self):
        return len(self._groups) if self._groups else 1

    def group_range(self, g):
        """[start, stop) env ids of group g"""
        return self._groups[g]["range"] if self._groups else (0, self.num_envs)

    def group_stream(self, g):
        """torch stream of group g: run that group's policy under `with torch.cuda.stream(env.group_stream(g))` and its
        steps need no cross-stream ordering at all"""
        return self._groups[g]["stream"]

    def step_group(self, g, actions):
        """one step of group g, enqueued on the group's stream.  actions [n_g,4] (a tensor produced on that stream, or
        ordered before it by the caller).  -> (obs [n_g,12], reward [n_g], done [n_g] bool, flags [n_g] u8, terminal_obs
        [n_g,12], terminal_state [n_g,26]) fresh tensors owned by the group's stream."""
        torch = _torch()
        G = self._groups[g]
        lo, hi = G["range"]
        n = hi - lo
        actions = self._as_device(actions, (n, 4))
        with torch.cuda.stream(G["stream"]):
            kw = dict(device=self.device)
            obs = torch.empty((n, self.obs_dim), dtype=torch.float32, **kw)
            rew = torch.empty((n,), dtype=torch.float32, **kw)
            done = torch.empty((n,), dtype=torch.uint8, **kw)
            flags = torch.empty((n,), dtype=torch.uint8, **kw)
            term = torch.empty((n, self.obs_dim), dtype=torch.float32, **kw) if self.auto_reset else None
            tst = torch.empty((n, 26), dtype=torch.float32, **kw) if (self.auto_reset and self._tstate is not None) else None
        _lib.check(self._lib.qs_step_group(self._h, g, self._ptr(actions), self._ptr(obs), self._ptr(rew), self._ptr(done),
                                           self._ptr(flags), self._ptr(term), self._ptr(tst)), "qs_step_group")
        self._nstep += 1
        return obs, rew, done.view(torch.bool), flags, term, tst

    def step_groups(self, actions):
        """one step of ALL groups in one call (G launches, one per group stream), full-batch tensors as step(); the
        outputs are the env's own buffers, complete once groups_join() / sync() has ordered the group streams"""
        self._actions = self._as_device(actions, (self.num_envs, 4))
        _lib.check(self._lib.qs_step_groups(self._h, self._ptr(self._actions), self._ptr(self._obs), self._ptr(self._rew),
                                            self._ptr(self._done), self._ptr(self._flags),
                                            self._ptr(self._term) if self.auto_reset else None,
                                            self._ptr(self._tstate) if self.auto_reset else None), "qs_step_groups")
        self._nstep += 1
        return self._obs, self._rew, self._done.view(_torch().bool)

    def groups_fork(self):
        """group streams wait for everything enqueued so far on the env's main stream (torch's current stream)"""
        self._use_current_stream()
        _lib.check(self._lib.qs_groups_fork(self._h), "qs_groups_fork")

    def groups_join(self):
        """the env's main stream waits for everything enqueued so far on the group streams"""
        self._use_current_stream()
        _lib.check(self._lib.qs_groups_join(self._h), "qs_groups_join")

    def rollout(self, actions=None, T=None, want_flags=True, stepwise=False, out=None):
        """T fused steps in one launch (qs_rollout), or T single-step launches issued from native code
        (stepwise=True, qs_rollout_stepwise).  actions [T,N,4] or None (in-kernel U(-1,1), fused only).
        out: optional (obs, reward, done, flags) tensors to write into.
        -> obs [T,N,12], reward [T,N], done [T,N] (uint8), flags [T,N] or None (torch tensors)."""
        torch = _torch()
        self._use_current_stream()
        if actions is not None:
            T = int(actions.shape[0])
            actions = self._as_device(actions, (T, self.num_envs, 4))
        elif T is None:
            raise ValueError("give actions or T")
        n = self.num_envs
        if out is not None:
            obs, rew, done, flags = out
        else:
            obs = torch.empty((T, n, self.obs_dim), dtype=torch.float32, device=self.device)
            rew = torch.empty((T, n), dtype=torch.float32, device=self.device)
            done = torch.empty((T, n), dtype=torch.uint8, device=self.device)
            flags = torch.empty((T, n), dtype=torch.uint8, device=self.device) if want_flags else None
        fn = self._lib.qs_rollout_stepwise if stepwise else self._lib.qs_rollout
        self._inputs_ready()
        if self._queue_host_ordered:
            _torch().cuda.current_stream(self.device).synchronize()
        _lib.check(fn(self._h, T, self._ptr(actions), self._ptr(obs), self._ptr(rew), self._ptr(done),
                      self._ptr(flags)), "qs_rollout")
        self._outputs_ready()
        if self._queue_host_ordered:
            self.sync()
        return obs, rew, done, flags

    def rollout_slab(self, actions=None, T=None, out=None):
        """qs_rollout_slab: T fused steps written as one packed slab [T,N,14] (obs 12, reward, done as 0/1)."""
        torch = _torch()
        self._use_current_stream()
        if actions is not None:
            T = int(actions.shape[0])
            actions = self._as_device(actions, (T, self.num_envs, 4))
        elif T is None:
            raise ValueError("give actions or T")
        slab = out if out is not None else torch.empty((T, self.num_envs, 14), dtype=torch.float32, device=self.device)
        self._inputs_ready()
        _lib.check(self._lib.qs_rollout_slab(self._h, T, self._ptr(actions), self._ptr(slab), None), "qs_rollout_slab")
        self._outputs_ready()
        return slab

    def random_actions(self, T, step0=None):
        """[T,N,4] synthetic U(-1,1) actions from the rocRAND action stream"""
        torch = _torch()
        if step0 is None:
            step0 = self.step_counter
        self._use_current_stream()
        a = torch.empty((T, self.num_envs, 4), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.qs_fill_random_actions(self._h, T, int(step0), self._ptr(a)), "qs_fill_random_actions")
        self._outputs_ready()
        return a

    def seed(self, seed=None):
        return [seed] * self.num_envs

    def render(self, mode="human"):
        return None

    # what stable_baselines' VecEnvWrapper machinery (VecNormalize, VecCheckNan, ...) asks of the wrapped VecEnv
    @property
    def unwrapped(self):
        return self

    def get_images(self):
        return []

    def getattr_depth_check(self, name, already_found):
        return None

    def _get_target_envs(self, indices):
        return [self for _ in (range(self.num_envs) if indices is None else indices)]

    def get_attr(self, name, indices=None):
        idx = range(self.num_envs) if indices is None else indices
        if name in ("state_chaser", "state_target"):
            st = self.get_state()
            arr = st["chaser" if name == "state_chaser" else "target"]
            return [arr[i] for i in idx]
        return [getattr(self, name) for _ in idx]

    def set_attr(self, name, value, indices=None):
        setattr(self, name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        idx = range(self.num_envs) if indices is None else indices
        return [getattr(self, method_name)(*args, **kwargs) for _ in idx]

    # ------------------------------------------------------------------ state / params
    @property
    def step_counter(self):
        k = C.c_uint64(0)
        _lib.check(self._lib.qs_get_step_counter(self._h, C.byref(k)), "qs_get_step_counter")
        return int(k.value)

    @step_counter.setter
    def step_counter(self, k):
        _lib.check(self._lib.qs_set_step_counter(self._h, int(k)), "qs_set_step_counter")

    def get_state(self, as_numpy=True):
        """dict(chaser[N,13], target[N,13], u_prev[N,8], qdes[N,4], last_shaping[N], t[N])"""
        torch = _torch()
        n = self.num_envs
        shapes = dict(chaser=(n, 13), target=(n, 13), u_prev=(n, 8), qdes=(n, 4), last_shaping=(n,), t=(n,))
        self._use_current_stream()
        bufs = {k: torch.empty(s, dtype=torch.float32, device=self.device) for k, s in shapes.items()}
        _lib.check(self._lib.qs_get_state(self._h, *[self._ptr(bufs[k]) for k in shapes]), "qs_get_state")
        self._outputs_ready()
        return {k: (v.cpu().numpy() if as_numpy else v) for k, v in bufs.items()}

    def set_state(self, chaser=None, target=None, u_prev=None, qdes=None, last_shaping=None, t=None):
        n = self.num_envs
        shapes = dict(chaser=(n, 13), target=(n, 13), u_prev=(n, 8), qdes=(n, 4), last_shaping=(n,), t=(n,))
        given = dict(chaser=chaser, target=target, u_prev=u_prev, qdes=qdes, last_shaping=last_shaping, t=t)
        self._use_current_stream()
        ptrs, keep = [], []
        for k, s in shapes.items():
            if given[k] is None:
                ptrs.append(None)
            else:
                tt = self._as_device(given[k], s)
                keep.append(tt)
                ptrs.append(self._ptr(tt))
        self._inputs_ready()
        _lib.check(self._lib.qs_set_state(self._h, *ptrs), "qs_set_state")
        self.sync()

    def set_init_state(self, chaser_init, target_init=None):
        """per-env initial states reset() returns to (env.chaser_ini_state / target_ini_state; HoveringEnv.ini_state)"""
        n = self.num_envs
        self._use_current_stream()
        c = self._as_device(chaser_init, (n, 13))
        t = self._as_device(target_init, (n, 13)) if target_init is not None else None
        self._inputs_ready()
        _lib.check(self._lib.qs_set_init_state(self._h, self._ptr(c), self._ptr(t)), "qs_set_init_state")

    def get_init_state(self):
        torch = _torch()
        n = self.num_envs
        self._use_current_stream()
        c = torch.empty((n, 13), dtype=torch.float32, device=self.device)
        t = torch.empty((n, 13), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.qs_get_init_state(self._h, self._ptr(c), self._ptr(t)), "qs_get_init_state")
        return c.cpu().numpy(), (None if self.kind == _lib.KIND_HOVER else t.cpu().numpy())

    def set_params(self, mass=None, inertia=None):
        n = self.num_envs
        self._use_current_stream()
        m = self._as_device(mass, (n,)) if mass is not None else None
        i = self._as_device(inertia, (n, 3)) if inertia is not None else None
        self._inputs_ready()
        _lib.check(self._lib.qs_set_params(self._h, self._ptr(m), self._ptr(i)), "qs_set_params")
        self.sync()

    def get_params(self):
        torch = _torch()
        n = self.num_envs
        self._use_current_stream()
        m = torch.empty((n,), dtype=torch.float32, device=self.device)
        i = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.qs_get_params(self._h, self._ptr(m), self._ptr(i)), "qs_get_params")
        self._outputs_ready()
        return m.cpu().numpy(), i.cpu().numpy()

    def timer_start(self):
        _lib.check(self._lib.qs_timer_start(self._h), "qs_timer_start")

    def timer_stop(self):
        ms = C.c_float(0)
        _lib.check(self._lib.qs_timer_stop(self._h, C.byref(ms)), "qs_timer_stop")
        return float(ms.value)


class InfoView:
    """infos[i] of the VecEnv protocol, materialised lazily: building N dicts per
    step would dominate the step at N = 65 536.  infos[i] is a dict with the
    reference's keys (docking_env.py:226-229) plus SB2's 'terminal_observation'.

    What belongs to THIS step is captured when step_wait() returns: done, flags, and -- for the envs that finished --
    the terminal observation and the terminal chaser / target states (written by the step kernel before its in-kernel
    reset: info['chaser'] is the state the reference's env returns on the done step, not the reset state).  The
    states of envs that did not finish are the env's current states: snapshotted per step with
    VecDockingEnv(info_state=True), else fetched on first access, which must come before the next step."""

    def __init__(self, env, done=None, flags=None, term=None, tstate=None, done_t=None, flags_t=None, term_t=None,
                 tstate_t=None):
        self._env = env
        self._done, self._flags, self._term, self._tstate = done, flags, term, tstate        # host arrays
        self._dev = (done_t, flags_t, term_t, tstate_t)                                       # this step's device tensors
        self._step = env._nstep
        self._state = None
        # copy=False: the env's buffers are rewritten by the next step; everything of this view is fetched on first access and
        # guarded by the step count (two per-step clones of done / flags cost 9 of the 16 us a step() call took on the host).
        # info_state=True asks for a per-step snapshot: then done / flags are snapshotted as well
        if env.backend == "torch" and not env.copy and env.info_state:
            self._dev = (done_t.clone(), flags_t.clone(), term_t, tstate_t)
        if env.info_state and env.kind != _lib.KIND_HOVER:
            st = env.get_state(as_numpy=False)
            self._state_dev = (st["chaser"], st["target"])
        else:
            self._state_dev = None

    def __len__(self):
        return self._env.num_envs

    def _stale(self):
        return self._env._nstep != self._step

    def _materialise(self):
        d, f, tm, ts = self._dev
        if self._done is None:
            if self._stale() and not self._env.copy and not self._env.info_state:
                raise _lib.QuadsimError("this InfoView is read after a later step of a copy=False env: its buffers were "
                                        "recycled; read infos before stepping again or use copy=True")
            self._done = d.cpu().numpy().astype(bool)
            self._flags = f.cpu().numpy()
        if self._term is None and self._env.auto_reset and self._done.any():
            if self._stale() and not self._env.copy:
                raise _lib.QuadsimError("this InfoView is read after a later step of a copy=False env: its terminal rows "
                                        "were recycled; read infos before stepping again or use copy=True")
            self._term = tm.cpu().numpy()
            self._tstate = ts.cpu().numpy() if ts is not None else None

    def _states(self):
        if self._state is None:
            if self._state_dev is not None:
                self._state = {"chaser": self._state_dev[0].cpu().numpy(), "target": self._state_dev[1].cpu().numpy()}
            else:
                if self._stale():
                    raise _lib.QuadsimError("info['chaser'/'target'] of an env that did not finish is the env's current state; "
                                            "this InfoView is read after a later step.  Read it before the next step or "
                                            "create the env with info_state=True")
                self._state = self._env.get_state()
        return self._state

    def __getitem__(self, i):
        self._materialise()
        f = int(self._flags[i])
        finished = bool(self._done[i]) and self._env.auto_reset
        if self._env.kind == _lib.KIND_HOVER:
            info = {}                                         # hovering_env.py:78 returns an empty info
        else:
            if finished:
                # docking_env.py:226-229: the states of the TERMINAL step (the in-kernel reset has already replaced them)
                chaser, target = self._tstate[i, :13].copy(), self._tstate[i, 13:].copy()
            else:
                st = self._states()
                chaser, target = st["chaser"][i], st["target"][i]
            info = {"chaser": chaser, "target": target,
                    "flag_docking": bool(f & _lib.FLAG_DOCKED), "done_overlimit": bool(f & _lib.FLAG_OVERLIMIT)}
        if finished:
            info["terminal_observation"] = self._term[i].copy()
        return info

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    @property
    def flags(self):
        self._materialise()
        return self._flags

    @property
    def terminal_states(self):
        """[N,26] chaser | target of the terminal step (rows of envs that did not finish are undefined); None if no env finished"""
        self._materialise()
        return self._tstate
