"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The env.step() path shards by env id with NO data-path collective (envs are
independent).  The only exchange step is the one BASELINE
configs 4/5 name: an all-gather of the roll-out slabs (obs, reward, done) so that
every rank holds the whole roll-out -- once per T-step roll-out, never per step.
xGMI is point-to-point (7 links x ~153 GB/s): three large gathers per roll-out,
not 3*T small ones.
"""
from .vec_env import shard_range


def env_shard(total_envs, rank=None, world_size=None):
    """(offset, count) of this rank's envs; rank/world default to the initialised process group"""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(total_envs, rank, world_size)
    return lo, hi - lo


def gather_rollout(obs, reward, done, out=None, group=None):
    """All-gather the per-rank roll-out slabs obs [T,n,12], reward [T,n], done [T,n] (equal n on every rank)
    into rank-major buffers [G,T,n,12], [G,T,n], [G,T,n].  Returns the three gathered tensors;
    `out` = (obs_all, reward_all, done_all) re-uses preallocated buffers.  `rollout_global_view` turns them
    into [T, G*n, ...] with the global env order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = (torch.empty((world,) + tuple(obs.shape), dtype=obs.dtype, device=obs.device),
               torch.empty((world,) + tuple(reward.shape), dtype=reward.dtype, device=reward.device),
               torch.empty((world,) + tuple(done.shape), dtype=done.dtype, device=done.device))
    # concatenation along dim 0 is the form every backend (RCCL and gloo) accepts
    for dst, src in zip(out, (obs, reward, done)):
        dist.all_gather_into_tensor(dst.view((world * src.shape[0],) + tuple(src.shape[1:])), src.contiguous(), group=group)
    return out


def gather_slab(slab, out=None, group=None):
    """ONE collective per roll-out: all-gather the packed slab [T,n,14] (obs 12, reward, done as 0/1 -- 56 B per
    env-step, written directly by qs_rollout_slab) into [G,T,n,14].  Fewer, larger collectives suit xGMI's
    point-to-point links (7 x ~153 GB/s): one 235 MB transfer per rank for T = 64, n = 65 536."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(slab.shape), dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(out.view((world * slab.shape[0],) + tuple(slab.shape[1:])), slab.contiguous(), group=group)
    return out


def split_slab(slab):
    """[..., 14] -> (obs [..., 12], reward [...], done [...] bool) views"""
    return slab[..., :12], slab[..., 12], slab[..., 13] > 0.5


def rollout_global_view(x_all):
    """[G,T,n,...] (rank-major) -> [T, G*n, ...]: contiguous env shards => global env id = g*n + i"""
    G, T, n = x_all.shape[:3]
    perm = (1, 0, 2) + tuple(range(3, x_all.dim()))
    return x_all.permute(*perm).reshape((T, G * n) + tuple(x_all.shape[3:]))
