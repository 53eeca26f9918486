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


class SlabGatherPipeline:
    """Roll-out k+1 overlapped with the all-gather of roll-out k ("overlap (double-buffer) or gather
    per rollout"): `depth` slab buffers rotate; `produce(slab)` fills one on the caller's stream (qs_rollout_slab), its
    all-gather is issued asynchronously (RCCL runs it on its own stream, ordered behind the producer) and the next roll-out
    starts at once.  `step()` returns the gathered [G,T,n,14] tensor of the roll-out submitted `depth` calls earlier (None
    while the pipeline fills); it stays valid -- for work enqueued on the caller's stream -- until the NEXT call of step()
    (depth + 1 output buffers rotate); `flush()` returns the ones still in flight, oldest first."""

    def __init__(self, produce, slab_shape, dtype=None, device=None, depth=2, group=None):
        import torch
        import torch.distributed as dist
        self._dist, self.produce, self.group, self.depth = dist, produce, group, max(1, int(depth))
        self.world = dist.get_world_size(group)
        dtype = dtype or torch.float32
        self.slabs = [torch.empty(tuple(slab_shape), dtype=dtype, device=device) for _ in range(self.depth)]
        self.outs = [torch.empty((self.world,) + tuple(slab_shape), dtype=dtype, device=device) for _ in range(self.depth + 1)]
        self.works = [None] * self.depth       # per slab: (work, index of its output buffer)
        self.k = 0

    def _wait(self, i):
        w, self.works[i] = self.works[i], None
        if w is None:
            return None
        w[0].wait()                            # the caller's stream (or the host, for CPU backends) is behind the gather
        return self.outs[w[1]]

    def step(self):
        i, o = self.k % self.depth, self.k % (self.depth + 1)
        ready = self._wait(i)                  # the gather that last read slab i
        s = self.slabs[i]
        self.produce(s)
        # output buffer o was handed out depth + 1 calls ago: its readers were enqueued before this call
        w = self._dist.all_gather_into_tensor(self.outs[o].view((self.world * s.shape[0],) + tuple(s.shape[1:])), s,
                                              group=self.group, async_op=True)
        self.works[i] = (w, o)
        self.k += 1
        return ready

    def flush(self):
        done = []
        for j in range(self.depth):
            r = self._wait((self.k + j) % self.depth)
            if r is not None:
                done.append(r)
        return done


def split_slab(slab):
    """[..., 14] -> (obs [..., 12], reward [...], done [...] bool) views"""
    return slab[..., :12], slab[..., 12], slab[..., 13] > 0.5


def rollout_global_view(x_all):
    """[G,T,n,...] (rank-major) -> [T, G*n, ...]: contiguous env shards => global env id = g*n + i"""
    G, T, n = x_all.shape[:3]
    perm = (1, 0, 2) + tuple(range(3, x_all.dim()))
    return x_all.permute(*perm).reshape((T, G * n) + tuple(x_all.shape[3:]))
