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


def fused_policy_rollout(env, policy, T, want_actions=True):
    """The same loop as rollout_with_policy in ONE kernel launch (qs_policy_rollout): MLP on the matrix cores
    (exact-f32 MFMA) + fused env step, T steps, no host involvement.  Starts from the envs' current state.
    Returns (obs [T,N,12], reward [T,N], done [T,N] u8, flags [T,N] u8, actions [T,N,4] or None)."""
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
    _lib.check(env._lib.qs_policy_rollout(env._h, T, *[p(w) for w in policy._wt], p(obs), p(rew), p(done), p(flags),
                                          p(acts)), "qs_policy_rollout")
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
