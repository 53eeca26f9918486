"""Policy-in-the-loop roll-outs on the device.

``MlpPolicy`` is the deterministic actor of the PPO2 model the reference ships
(`trained_model/best_model_v0.zip`: shared_fc0 12->128, pi_fc0 128->128, pi 128->4,
ReLU, actions clipped to [-1, 1]) as `run_trained_docking_ppo2.py:37-45` uses it:
``action, _ = model.predict(obs, deterministic=True); env.step(action)``.
The three GEMMs are plain library GEMMs (torch -> hipBLASLt); the env step is the
fused HIP kernel.  The loop never leaves the GPU: no host sync between steps.
"""
import numpy as np


class MlpPolicy:
    def __init__(self, weights, device="cuda"):
        import torch
        self.torch = torch
        g = lambda k: torch.as_tensor(np.asarray(weights[k], np.float32)).to(device)  # noqa: E731
        self.w0, self.b0 = g("w0"), g("b0")
        self.w1, self.b1 = g("w1"), g("b1")
        self.w2, self.b2 = g("w2"), g("b2")

    @classmethod
    def from_npz(cls, path, device="cuda"):
        with np.load(path, allow_pickle=False) as z:
            return cls({k: z[k] for k in z.files}, device)

    def predict(self, obs):
        """obs [N,12] float32 (device) -> actions [N,4] in [-1,1]"""
        t = self.torch
        h = t.relu(t.addmm(self.b0, obs, self.w0))
        h = t.relu(t.addmm(self.b1, h, self.w1))
        return t.clamp(t.addmm(self.b2, h, self.w2), -1.0, 1.0)


    def predict_hip(self, env, obs, precision="f32", out=None):
        """the same actor as ONE hand-written kernel on the matrix cores (qs_policy_forward: exact-f32 MFMA; "bf16x3":
        split-bf16 operands, ~1e-5 on an action), launched on `env`'s stream: obs [n,12] float32 device tensor -> actions
        [n,4].  65 536 rows: ~22 us against ~80 us for the three library GEMMs of predict()."""
        import ctypes as C
        import torch
        from . import _lib
        if (not isinstance(obs, torch.Tensor) or obs.dtype != torch.float32 or obs.dim() != 2 or obs.shape[1] != 12
                or obs.device != env.device):
            raise ValueError("predict_hip: obs must be a float32 [n, 12] tensor on %s" % (env.device,))
        obs = obs.contiguous()
        n = int(obs.shape[0])
        if out is not None and (out.dtype != torch.float32 or tuple(out.shape) != (n, 4) or not out.is_contiguous() or out.device != obs.device):
            raise ValueError("predict_hip: out must be a contiguous float32 [n, 4] tensor on the same device")
        acts = out if out is not None else torch.empty((n, 4), dtype=torch.float32, device=obs.device)
        p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
        env._use_current_stream()
        if precision == "bf16x3":
            if not hasattr(self, "_blob"):
                blob = pack_fast_weights(self)
                assert blob.size == env._lib.qs_policy_rollout_fast_blob_bytes()
                self._blob = torch.as_tensor(blob.copy()).to(obs.device)
            _lib.check(env._lib.qs_policy_forward_fast(env._h, n, p(self._blob), p(obs), p(acts)), "qs_policy_forward_fast")
        elif precision == "f32":
            if not hasattr(self, "_wt"):
                self._wt = [self.w0.t().contiguous(), self.b0.contiguous(), self.w1.t().contiguous(),
                            self.b1.contiguous(), self.w2.t().contiguous(), self.b2.contiguous()]
            _lib.check(env._lib.qs_policy_forward(env._h, n, *[p(w) for w in self._wt], p(obs), p(acts)), "qs_policy_forward")
        else:
            raise ValueError("precision must be 'f32' or 'bf16x3'")
        return acts


def _bf16_bits(x):
    """float32 -> bfloat16 bit pattern, round to nearest even (what v_cvt_pk_bf16_f32 does)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7FFF)) >> 16).astype(np.uint16)


def _bf16_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def pack_fast_weights(policy):
    """Weight image of qs_policy_rollout_fast: every weight split into bf16 hi + lo, A-operand fragments stored
    ready-made in the k-order the accumulator-as-B-operand chaining needs (csrc/policy_rollout.hpp, 'Fast actor')."""
    w1t = policy.w0.t().contiguous().cpu().numpy()      # [128][12]
    w2t = policy.w1.t().contiguous().cpu().numpy()      # [128][128]
    w3t = policy.w2.t().contiguous().cpu().numpy()      # [4][128]
    lane = np.arange(64); g, c = lane >> 4, lane & 15
    j = np.arange(8)
    hid = lambda p: 16 * (2 * p + (j[None, :] >> 2)) + 4 * g[:, None] + (j[None, :] & 3)       # noqa: E731  [lane][j]
    a2 = np.zeros((8, 4, 64, 8), np.float32)
    for nt in range(8):
        for p in range(4):
            a2[nt, p] = w2t[(16 * nt + c)[:, None], hid(p)]
    a1 = np.zeros((8, 64, 8), np.float32)
    k1 = 8 * g[:, None] + j[None, :]
    for rt in range(8):
        a1[rt] = np.where(k1 < 12, w1t[(16 * rt + c)[:, None], np.minimum(k1, 11)], 0.0)
    a3 = np.zeros((4, 64, 8), np.float32)
    for q in range(4):
        a3[q] = np.where((c < 4)[:, None], w3t[np.minimum(c, 3)[:, None], hid(q)], 0.0)
    parts = []
    for a in (a2, a1, a3):
        hi = _bf16_bits(a)
        lo = _bf16_bits(a - _bf16_to_f32(hi))
        parts += [hi.tobytes(), lo.tobytes()]
    b3 = np.zeros(16, np.float32); b3[:4] = policy.b2.cpu().numpy()
    parts += [policy.b0.cpu().numpy().astype(np.float32).tobytes(), policy.b1.cpu().numpy().astype(np.float32).tobytes(),
              b3.tobytes()]
    blob = np.frombuffer(b"".join(parts), np.uint8)
    return blob


def fused_policy_rollout(env, policy, T, want_actions=True, precision="f32"):
    """The same loop as rollout_with_policy in ONE kernel launch: MLP on the matrix cores + fused env step, T steps,
    no host involvement.  Starts from the envs' current state.
    precision "f32": qs_policy_rollout, exact-float32 MFMA (an ordinary float32 network evaluation);
    precision "bf16x3": qs_policy_rollout_fast, split-bf16 operands on the 16x faster bf16 matrix rate, ~1e-5 error
    on the actions.  Returns (obs [T,N,12], reward [T,N], done [T,N] u8, flags [T,N] u8, actions [T,N,4] or None)."""
    import ctypes as C
    import torch
    from . import _lib
    if not hasattr(policy, "_wt"):
        policy._wt = [policy.w0.t().contiguous(), policy.b0.contiguous(), policy.w1.t().contiguous(),
                      policy.b1.contiguous(), policy.w2.t().contiguous(), policy.b2.contiguous()]
    n, dev = env.num_envs, env.device
    obs = torch.empty((T, n, 12), dtype=torch.float32, device=dev)
    rew = torch.empty((T, n), dtype=torch.float32, device=dev)
    done = torch.empty((T, n), dtype=torch.uint8, device=dev)
    flags = torch.empty((T, n), dtype=torch.uint8, device=dev)
    acts = torch.empty((T, n, 4), dtype=torch.float32, device=dev) if want_actions else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    env._use_current_stream()
    if precision == "bf16x3":
        if not hasattr(policy, "_blob"):
            blob = pack_fast_weights(policy)
            assert blob.size == env._lib.qs_policy_rollout_fast_blob_bytes()
            policy._blob = torch.as_tensor(blob.copy()).to(dev)
        _lib.check(env._lib.qs_policy_rollout_fast(env._h, T, p(policy._blob), p(obs), p(rew), p(done), p(flags), p(acts)),
                   "qs_policy_rollout_fast")
    elif precision == "f32":
        _lib.check(env._lib.qs_policy_rollout(env._h, T, *[p(w) for w in policy._wt], p(obs), p(rew), p(done), p(flags),
                                              p(acts)), "qs_policy_rollout")
    else:
        raise ValueError("precision must be 'f32' or 'bf16x3'")
    return obs, rew, done, flags, acts


def rollout_with_policy(env, policy, T, obs0=None):
    """T steps of ``a = policy(obs); obs, r, d, info = env.step(a)`` on a torch-backend VecDockingEnv.
    Returns stacked (obs [T,N,12], reward [T,N], done [T,N] bool, flags [T,N] u8, actions [T,N,4])."""
    import torch
    obs = env.reset() if obs0 is None else obs0
    O, R, D, F, A = [], [], [], [], []
    for _ in range(T):
        a = policy.predict(obs)
        obs, r, d, info = env.step(a)
        O.append(obs.clone()); R.append(r.clone()); D.append(d.clone()); F.append(env._flags.clone()); A.append(a)
    return torch.stack(O), torch.stack(R), torch.stack(D), torch.stack(F), torch.stack(A)
