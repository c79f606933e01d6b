"""Batch sharding across the GPUs of one node.

Environments never interact, so the batch partitions into contiguous env ranges,
one process per GPU, with NO collective inside the physics step.  The only
exchanges are at the agent interface: actions (B, nu) scattered from the learner
rank and observations / rewards gathered back -- `torch.distributed` over RCCL
(`backend='nccl'` on ROCm) on the GPU box, `gloo` in CPU tests.
"""
import numpy as np


def shard_bounds(batch_size, world_size, rank):
  """Contiguous [lo, hi) of rank's environments; sizes differ by at most one."""
  if not 0 <= rank < world_size:
    raise ValueError('rank out of range')
  base, rem = divmod(batch_size, world_size)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(batch_size, world_size):
  return [shard_bounds(batch_size, world_size, r)[1] - shard_bounds(batch_size, world_size, r)[0]
          for r in range(world_size)]


class ShardedEnvBatch:
  """Agent-interface collectives for an env-sharded batch (one instance per rank)."""

  def __init__(self, global_batch, dist=None, device='cpu'):
    import torch
    self._torch = torch
    self.dist = dist
    self.world = dist.get_world_size() if dist is not None else 1
    self.rank = dist.get_rank() if dist is not None else 0
    self.global_batch = global_batch
    self.lo, self.hi = shard_bounds(global_batch, self.world, self.rank)
    self.sizes = shard_sizes(global_batch, self.world)
    self.device = device

  @property
  def local_batch(self):
    return self.hi - self.lo

  def scatter_actions(self, actions_global, src=0):
    """actions_global: (B, nu) tensor on rank `src` (ignored elsewhere) ->
    this rank's (local_batch, nu) slice."""
    torch = self._torch
    if self.world == 1:
      return actions_global[self.lo:self.hi]
    nu = torch.tensor([actions_global.shape[1] if self.rank == src else 0], device=self.device)
    self.dist.broadcast(nu, src)
    pad = max(self.sizes)
    out = torch.empty((pad, int(nu.item())), dtype=torch.float32, device=self.device)
    chunks = None
    if self.rank == src:
      chunks = []
      for r in range(self.world):
        lo, hi = shard_bounds(self.global_batch, self.world, r)
        c = torch.zeros((pad, int(nu.item())), dtype=torch.float32, device=self.device)
        c[:hi - lo] = actions_global[lo:hi].to(torch.float32)
        chunks.append(c)
    self.dist.scatter(out, chunks, src=src)
    return out[:self.local_batch]

  def gather(self, local, dst=None):
    """local: (local_batch, n) tensor -> (B, n) on every rank (all_gather)."""
    torch = self._torch
    if self.world == 1:
      return local
    pad = max(self.sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=self.device)
    buf[:self.local_batch] = local
    outs = [torch.empty_like(buf) for _ in range(self.world)]
    self.dist.all_gather(outs, buf)
    return torch.cat([o[:n] for o, n in zip(outs, self.sizes)], dim=0)

  def max_over_ranks(self, value):
    """Slowest rank's elapsed time (bench.py contract)."""
    torch = self._torch
    if self.world == 1:
      return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
    return float(t.item())
