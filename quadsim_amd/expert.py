"""PID expert + expert-dataset writer.

``PIDExpert`` is the scripted docking policy of run_expert_policy.py:49-69: a velocity controller that flies the
chaser to 0.2 m behind the target.  ``record_expert_dataset`` is run_expert_record.py:121-189 for N parallel envs:
it returns / saves the SB2 ``ExpertDataset`` dictionary (keys actions, obs, rewards, episode_returns,
episode_starts) that run_pretrained_ppo2_docking.py:50-69 feeds to behaviour cloning and GAIL.
"""
import ctypes as C

import numpy as np

from . import _lib


class PIDExpert:
    def __init__(self, env, kp=0.35, kd=0.0):
        import torch
        self.env, self.kp, self.kd = env, float(kp), float(kd)
        n = env.num_envs
        try:
            c, _ = env.get_init_state()                     # stored initial states (docking-v1 / set_init_state)
        except _lib.QuadsimError:
            c = np.tile(np.array([8, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], np.float32), (n, 1))
        self.state_des = torch.as_tensor(np.ascontiguousarray(c, np.float32)).to(env.device)   # = env.chaser_ini_state (:44)
        self._actions = torch.empty((n, 4), dtype=torch.float32, device=env.device)

    def act(self):
        """expert actions [N,4] for the envs' current states (a fresh tensor view is overwritten by the next call)"""
        e = self.env
        _lib.check(e._lib.qs_expert_action(e._h, C.c_void_p(self.state_des.data_ptr()), self.kp, self.kd,
                                           C.c_void_p(self._actions.data_ptr())), "qs_expert_action")
        return self._actions


def record_expert_dataset(env, n_steps, expert=None, save_path=None):
    """Roll the expert for n_steps in every env (auto-reset on) and return the ExpertDataset dict with the
    env-major flattening the single-env recorder produces (each env's time series is contiguous)."""
    import torch
    from .rollout_buffer import swap_and_flatten
    expert = expert or PIDExpert(env)
    n = env.num_envs
    obs = env.reset()
    O, A, R, D = [], [], [], []
    for _ in range(n_steps):
        a = expert.act()
        O.append(obs.clone()); A.append(a.clone())
        obs, r, d, _ = env.step(a)
        R.append(r.clone()); D.append(d.clone())
    O, A, R, D = torch.stack(O), torch.stack(A), torch.stack(R), torch.stack(D)
    starts = torch.ones((n_steps, n), dtype=torch.bool, device=env.device)
    starts[1:] = D[:-1]                                       # episode_starts.append(done) shifted by one (:111,:146)
    ret = torch.zeros(n, device=env.device)
    rets = []
    Rn, Dn = R.cpu().numpy(), D.cpu().numpy()
    acc = np.zeros(n)
    for t in range(n_steps):                                 # episode_returns in env-major order, like the recorder
        acc += Rn[t]
        for i in np.nonzero(Dn[t])[0]:
            rets.append((i, t, acc[i])); acc[i] = 0.0
    rets.sort()
    data = {
        "actions": swap_and_flatten(env, A).cpu().numpy(),
        "obs": swap_and_flatten(env, O).cpu().numpy(),
        "rewards": swap_and_flatten(env, R).cpu().numpy(),
        "episode_returns": np.array([x[2] for x in rets]),
        "episode_starts": starts.cpu().numpy().swapaxes(0, 1).reshape(-1),
    }
    del ret
    if save_path is not None:
        np.savez(save_path, **data)
    return data
