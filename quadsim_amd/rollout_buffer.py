"""On-device roll-out post-processing: the part of the in-tree PPO2 Runner that sits
either side of ``env.step`` -- GAE(lambda) (rl_baselines/ppo2/ppo2.py:507-520) and ``swap_and_flatten``
(:531-539) -- on the [T,N,.] tensors a roll-out leaves in HBM.  Same names and argument meaning as the reference."""
import ctypes as C

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def compute_gae(env, rewards, values, dones, last_values, last_dones, gamma=0.99, lam=0.95):
    """mb_rewards, mb_values [T,N] float32, mb_dones [T,N] (done BEFORE step t), last_values [N], last dones [N]
    -> (mb_advs [T,N], mb_returns [T,N]) as torch tensors on the env's device."""
    import torch
    T, n = rewards.shape
    dev = env.device
    f = lambda x: x.to(device=dev, dtype=torch.float32).contiguous()          # noqa: E731
    b = lambda x: x.to(device=dev).to(torch.uint8).contiguous()                # noqa: E731
    rewards, values, last_values = f(rewards), f(values), f(last_values)
    dones, last_dones = b(dones), b(last_dones)
    advs = torch.empty((T, n), dtype=torch.float32, device=dev)
    rets = torch.empty((T, n), dtype=torch.float32, device=dev)
    _lib.check(env._lib.qs_gae(env._h, T, n, _ptr(rewards), _ptr(values), _ptr(dones), _ptr(last_values),
                               _ptr(last_dones), float(gamma), float(lam), _ptr(advs), _ptr(rets)), "qs_gae")
    return advs, rets


def swap_and_flatten(env, arr):
    """[T,N,...] -> [N*T,...] (env-major), as rl_baselines/ppo2/ppo2.py:531-539"""
    import torch
    T, n = arr.shape[0], arr.shape[1]
    d = 1
    for k in arr.shape[2:]:
        d *= int(k)
    x = arr.to(device=env.device, dtype=torch.float32).contiguous()
    out = torch.empty((n * T,) + tuple(arr.shape[2:]), dtype=torch.float32, device=env.device)
    _lib.check(env._lib.qs_swap_and_flatten(env._h, T, n, d, _ptr(x), _ptr(out)), "qs_swap_and_flatten")
    return out
