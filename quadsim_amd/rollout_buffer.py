"""On-device roll-out post-processing: the part of the in-tree PPO2 Runner that sits
either side of ``env.step`` -- GAE(lambda) (rl_baselines/ppo2/ppo2.py:507-520), ``swap_and_flatten``
(:531-539) and the episode accounting behind ``ep_infos`` (:486-489) -- on the [T,N,.] tensors a roll-out leaves
in HBM.  Same names and argument meaning as the reference.  Every launch goes to the stream torch considers
current (the env follows it), so the torch temporaries made here and the kernels that read them stay ordered."""
import ctypes as C

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f32(env, x):
    import torch
    return x.to(device=env.device, dtype=torch.float32).contiguous()


def _u8(env, x):
    import torch
    x = x.to(device=env.device)
    return (x if x.dtype == torch.uint8 else x.to(torch.uint8)).contiguous()


def compute_gae(env, rewards, values, dones, last_values, last_dones, gamma=0.99, lam=0.95):
    """mb_rewards, mb_values [T,N] float32, mb_dones [T,N] (done BEFORE step t), last_values [N], last dones [N]
    -> (mb_advs [T,N], mb_returns [T,N]) as torch tensors on the env's device."""
    import torch
    T, n = rewards.shape
    dev = env.device
    env._use_current_stream()
    rewards, values, last_values = _f32(env, rewards), _f32(env, values), _f32(env, last_values)
    dones, last_dones = _u8(env, dones), _u8(env, last_dones)
    advs = torch.empty((T, n), dtype=torch.float32, device=dev)
    rets = torch.empty((T, n), dtype=torch.float32, device=dev)
    env._inputs_ready()
    _lib.check(env._lib.qs_gae(env._h, T, n, _ptr(rewards), _ptr(values), _ptr(dones), _ptr(last_values),
                               _ptr(last_dones), float(gamma), float(lam), _ptr(advs), _ptr(rets)), "qs_gae")
    env._outputs_ready()
    return advs, rets


def swap_and_flatten(env, arr):
    """[T,N,...] -> [N*T,...] (env-major), as rl_baselines/ppo2/ppo2.py:531-539.  float32 in, float32 out;
    uint8 / bool in, the same dtype out (no float round trip)."""
    import torch
    T, n = arr.shape[0], arr.shape[1]
    d = 1
    for k in arr.shape[2:]:
        d *= int(k)
    env._use_current_stream()
    if arr.dtype in (torch.uint8, torch.bool) and d == 1:
        x = arr.to(device=env.device).contiguous()
        out = torch.empty((n * T,) + tuple(arr.shape[2:]), dtype=x.dtype, device=env.device)
        env._inputs_ready()
        _lib.check(env._lib.qs_swap_and_flatten_u8(env._h, T, n, _ptr(x), _ptr(out)), "qs_swap_and_flatten_u8")
        env._outputs_ready()
        return out
    x = _f32(env, arr)
    out = torch.empty((n * T,) + tuple(arr.shape[2:]), dtype=torch.float32, device=env.device)
    env._inputs_ready()
    _lib.check(env._lib.qs_swap_and_flatten(env._h, T, n, d, _ptr(x), _ptr(out)), "qs_swap_and_flatten")
    env._outputs_ready()
    return out


def gae_and_flatten(env, rewards, values, neglogp, dones, last_values, last_dones, gamma=0.99, lam=0.95, want_advs=False):
    """GAE (ppo2.py:507-520) and the env-major flatten (:522-523) of every per-(t, env) scalar of the roll-out in ONE
    pass (qs_gae_flatten).  -> dict: returns, values, neglogp (None if not given), rewards [N*T] float32, masks [N*T]
    bool (= mb_dones); with want_advs also advs / returns_tm [T,N] time-major."""
    import torch
    T, n = rewards.shape
    dev = env.device
    env._use_current_stream()
    rewards, values, last_values = _f32(env, rewards), _f32(env, values), _f32(env, last_values)
    neglogp = _f32(env, neglogp) if neglogp is not None else None
    dones, last_dones = _u8(env, dones), _u8(env, last_dones)
    mk = lambda dt=torch.float32: torch.empty((n * T,), dtype=dt, device=dev)      # noqa: E731
    out = {"returns": mk(), "values": mk(), "neglogp": mk() if neglogp is not None else None, "rewards": mk(),
           "masks": mk(torch.uint8)}
    advs = torch.empty((T, n), dtype=torch.float32, device=dev) if want_advs else None
    rets = torch.empty((T, n), dtype=torch.float32, device=dev) if want_advs else None
    env._inputs_ready()
    _lib.check(env._lib.qs_gae_flatten(env._h, T, n, _ptr(rewards), _ptr(values), _ptr(neglogp), _ptr(dones),
                                       _ptr(last_values), _ptr(last_dones), float(gamma), float(lam),
                                       _ptr(out["returns"]), _ptr(out["values"]), _ptr(out["neglogp"]), _ptr(out["rewards"]),
                                       _ptr(out["masks"]), _ptr(advs), _ptr(rets)), "qs_gae_flatten")
    env._outputs_ready()
    out["masks"] = out["masks"].view(torch.bool)
    out["advs"], out["returns_tm"] = advs, rets
    return out


class EpisodeTracker:
    """Returns and lengths of the episodes that end inside each roll-out (qs_episode_stats): what the Monitor wrapper
    reports as info['episode'] and Runner._run collects into ep_infos (ppo2.py:486-489; run_docking_ppo2.py:19-35).
    The unfinished episode of every env is carried from one roll-out to the next on the device."""

    def __init__(self, env):
        import torch
        self.env = env
        n = env.num_envs
        self.ep_ret = torch.zeros((n,), dtype=torch.float32, device=env.device)
        self.ep_len = torch.zeros((n,), dtype=torch.int32, device=env.device)
        self._count = torch.zeros((1,), dtype=torch.int64, device=env.device)
        self._bufs = None
        self._n = 0                                    # no roll-out seen yet: no episode has ended

    def update(self, rewards, dones, last_dones):
        """rewards [T,N]; dones [T,N] u8 (flags BEFORE each step); last_dones [N] u8 -> self (lazy results)"""
        import torch
        env = self.env
        T, n = rewards.shape
        env._use_current_stream()
        rewards, dones, last_dones = _f32(env, rewards), _u8(env, dones), _u8(env, last_dones)
        cap = T * n                                    # an episode is at least one step long: cannot overflow
        if self._bufs is None or self._bufs[0].numel() < cap:
            self._bufs = (torch.empty((cap,), dtype=torch.int64, device=env.device),
                          torch.empty((cap,), dtype=torch.float32, device=env.device),
                          torch.empty((cap,), dtype=torch.int32, device=env.device))
        key, ret, ln = self._bufs
        env._inputs_ready()
        _lib.check(env._lib.qs_episode_stats(env._h, T, n, _ptr(rewards), _ptr(dones), _ptr(last_dones), _ptr(self.ep_ret),
                                             _ptr(self.ep_len), _ptr(self._count), cap, _ptr(key), _ptr(ret), _ptr(ln)),
                   "qs_episode_stats")
        env._outputs_ready()
        self._n = None
        return self

    @property
    def count(self):
        """episodes that ended in the last roll-out (one device -> host read, cached)"""
        if self._n is None:
            self._n = int(self._count.item())
        return self._n

    def results(self, ordered=True):
        """(returns [E] float32, lengths [E] int32, keys [E] int64 = t*N + env) device tensors of the last roll-out;
        ordered: sorted by key = the reference's (step, env) order"""
        import torch
        e = self.count
        if self._bufs is None:                         # queried before the first update()
            dev = self.env.device
            return (torch.empty((0,), dtype=torch.float32, device=dev), torch.empty((0,), dtype=torch.int32, device=dev),
                    torch.empty((0,), dtype=torch.int64, device=dev))
        key, ret, ln = (b[:e] for b in self._bufs)
        if ordered and e > 1:
            key, order = torch.sort(key)
            ret, ln = ret[order], ln[order]
        return ret, ln, key
