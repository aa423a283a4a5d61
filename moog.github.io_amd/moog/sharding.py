"""Env-axis sharding: env instances are independent, so the N-GPU path
is a static partition of the env axis with no data-path collective.

`shard_range(total, rank, world)` gives the contiguous block a rank owns; the
per-env RNG key is the *global* env index (`env_index0 + local`), so results do
not depend on how the axis is split.  `max_over_ranks` is the only collective
used, and only by the benchmark's timing (off the hot path).
"""


def shard_range(total_envs, rank, world_size):
    """Contiguous, near-equal blocks: rank r owns [start, start + count)."""
    if not 0 <= rank < world_size:
        raise ValueError('rank %d outside world of %d' % (rank, world_size))
    base, extra = divmod(int(total_envs), int(world_size))
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def max_over_ranks(value, device=None):
    """MAX all-reduce of a Python float over the default process group (RCCL on
    GPUs via backend 'nccl', gloo on CPU); identity when not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_over_ranks(values, op='sum', device=None):
    """SUM / MIN / MAX all-reduce of a list of Python floats over the default process group; identity when there is
    none.  The benchmark's bookkeeping (ranks that really took part, slowest / fastest rank) -- never the data path."""
    import torch
    import torch.distributed as dist
    vals = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return vals
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    dist.all_reduce(t, op={'sum': dist.ReduceOp.SUM, 'min': dist.ReduceOp.MIN, 'max': dist.ReduceOp.MAX}[op])
    return [float(x) for x in t.tolist()]


class MultiDeviceEnvironment(object):
    """One process driving several GPUs: one engine handle + stream per device,
    each owning a contiguous env block.  Outputs stay device-local (lists of
    per-device TimeSteps); `gather()` concatenates on the host for API parity."""

    def __init__(self, num_envs, devices, seed=0, **config):
        from . import environment
        self.devices = list(devices)
        self.shards = []
        for r, dev in enumerate(self.devices):
            start, count = shard_range(num_envs, r, len(self.devices))
            self.shards.append(environment.BatchedEnvironment(
                num_envs=count, device=dev, seed=seed, env_index0=start, **config))
        self.num_envs = num_envs

    def reset(self):
        return [s.reset() for s in self.shards]

    def step(self, actions):
        """actions: list of per-device tensors, or one host array split by shard."""
        if not isinstance(actions, (list, tuple)):
            out, off = [], 0
            for s in self.shards:
                out.append(actions[off:off + s.num_envs])
                off += s.num_envs
            actions = out
        return [s.step(a) for s, a in zip(self.shards, actions)]   # launches are asynchronous

    @staticmethod
    def gather(timesteps):
        import torch
        from . import _dm_env as dm_env
        cat = lambda xs: torch.cat([x.cpu() for x in xs])
        obs = {k: cat([t.observation[k] for t in timesteps]) for k in timesteps[0].observation}
        return dm_env.TimeStep(cat([t.step_type for t in timesteps]), cat([t.reward for t in timesteps]),
                               cat([t.discount for t in timesteps]), obs)
