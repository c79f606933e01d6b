"""Batch sharding across the GPUs of one node.

Environments never interact, so the batch partitions into contiguous env ranges,
one process per GPU, with NO collective inside the physics step.  The only
exchanges are at the agent interface: actions (B, nu) scattered from the learner
rank and observations / rewards gathered back -- `torch.distributed` over RCCL
(`backend='nccl'` on ROCm) on the GPU box, `gloo` in CPU tests.

`ShardedEnvBatch` issues the exchange in line with the step (what `bench.py`'s `collectives` leg prices as
"serialised").  `PipelinedExchange` is the overlap-ready form: every rank's shard is stepped as P part-batches (P = 2:
double buffering) with one set of exchange buffers and one pending collective per part, so that the all-gather of part
p's observations and the scatter of its next actions are in flight while part p + 1's step launch runs -- an on-policy
loop cannot overlap an environment's exchange with ITS OWN next step (a_{t+1} = pi(o_t)), it can with the other half's.
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
  """Agent-interface collectives for an env-sharded batch (one instance per rank).

  Per-step calls allocate nothing: the padded staging / receive buffers are created on the first call for a given
  (width, dtype) and reused; when every rank owns the same number of environments (the BASELINE shardings:
  4096 or 256 per GPU) the collectives work directly on views of the caller's tensors -- `scatter` from row
  blocks of the global action matrix, `all_gather_into_tensor` into one (B, n) result buffer."""

  def __init__(self, global_batch, dist=None, device='cpu'):
    import torch
    self._torch = torch
    self.dist = dist
    self.world = dist.get_world_size() if dist is not None else 1
    self.rank = dist.get_rank() if dist is not None else 0
    self.global_batch = global_batch
    self.lo, self.hi = shard_bounds(global_batch, self.world, self.rank)
    self.sizes = shard_sizes(global_batch, self.world)
    self.bounds = [shard_bounds(global_batch, self.world, r) for r in range(self.world)]
    self.even = len(set(self.sizes)) == 1
    self.pad = max(self.sizes)
    self.device = device
    self._buffers = {}
    self._nu = None

  @property
  def local_batch(self):
    return self.hi - self.lo

  def _buffer(self, tag, shape, dtype):
    key = (tag, tuple(shape), dtype)
    b = self._buffers.get(key)
    if b is None:
      b = self._buffers[key] = self._torch.zeros(tuple(shape), dtype=dtype, device=self.device)
    return b

  def scatter_actions(self, actions_global, src=0):
    """actions_global: (B, nu) tensor on rank `src` (ignored elsewhere) -> this rank's (local_batch, nu) slice
    (a view of a persistent receive buffer: valid until the next call)."""
    torch = self._torch
    if self.world == 1:
      return actions_global[self.lo:self.hi]
    if self._nu is None:        # once: the action width travels with the first exchange
      nu = torch.tensor([actions_global.shape[1] if self.rank == src else 0], device=self.device)
      self.dist.broadcast(nu, src)
      self._nu = int(nu.item())
    nu = self._nu
    out = self._buffer('act_recv', (self.pad, nu), torch.float32)
    chunks = None
    if self.rank == src:
      a = actions_global
      if a.dtype != torch.float32 or not a.is_contiguous():
        stage = self._buffer('act_cast', (self.global_batch, nu), torch.float32)
        stage.copy_(a)
        a = stage
      if self.even:
        chunks = [a[lo:hi] for lo, hi in self.bounds]                 # views, no copy
      else:
        stage = self._buffer('act_stage', (self.world, self.pad, nu), torch.float32)
        for r, (lo, hi) in enumerate(self.bounds):
          stage[r, :hi - lo].copy_(a[lo:hi])
        chunks = list(stage.unbind(0))
    self.dist.scatter(out, chunks, src=src)
    return out[:self.local_batch]

  def gather(self, local, dst=None):
    """local: (local_batch, n) tensor -> (B, n) on every rank (one all_gather_into_tensor; the result is a
    persistent buffer, valid until the next call with the same shape)."""
    torch = self._torch
    if self.world == 1:
      return local
    tail = tuple(local.shape[1:])
    recv = self._buffer('obs_recv', (self.world * self.pad,) + tail, local.dtype)
    if self.even:
      send = local if local.is_contiguous() else local.contiguous()
    else:
      send = self._buffer('obs_send', (self.pad,) + tail, local.dtype)
      send[:self.local_batch].copy_(local)
    self.dist.all_gather_into_tensor(recv, send)
    if self.even:
      return recv
    out = self._buffer('obs_out', (self.global_batch,) + tail, local.dtype)
    for r, (lo, hi) in enumerate(self.bounds):
      out[lo:hi].copy_(recv[r * self.pad:r * self.pad + hi - lo])
    return out

  def max_over_ranks(self, value):
    """Slowest rank's elapsed time (bench.py contract)."""
    torch = self._torch
    if self.world == 1:
      return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
    return float(t.item())


class PipelinedExchange:
  """Double-buffered agent-interface exchange for P part-batches per rank (see the module docstring).

  Part p of rank r owns the environments `part_bounds[p]` of r's shard; globally, part p is the union of every rank's
  part p, ordered by rank -- `gather_wait(p)` returns that (B_p, n) matrix and `scatter_async(p, actions)` takes one of the
  same row order.  Collectives are issued with `async_op=True`: with NCCL / RCCL they are enqueued on the communicator's
  own stream behind whatever the current stream has produced, and `*_wait` makes the current stream (not the host) wait
  for them; with gloo they run on its worker threads.  The order of the calls is the overlap:

      for p in parts:  a = scatter_wait(p); step(p, a); gather_async(p, observe(p))     # part p + 1's launch is enqueued
      for p in parts:  o = gather_wait(p); scatter_async(p, policy(o))                   # behind part p's all-gather

  Every environment sees exactly the operations, in exactly the order, of the serial loop (scatter, step, gather) -- the
  parts only interleave -- so trajectories are bit-equal to `ShardedEnvBatch`'s (tests/test_sharding_gloo.py)."""

  def __init__(self, global_batch, parts=2, dist=None, device='cpu'):
    import torch
    self._torch = torch
    self.dist = dist
    self.world = dist.get_world_size() if dist is not None else 1
    self.rank = dist.get_rank() if dist is not None else 0
    self.parts = int(parts)
    self.device = device
    self.lo, self.hi = shard_bounds(global_batch, self.world, self.rank)
    local = self.hi - self.lo
    sizes = set(shard_sizes(global_batch, self.world))
    if len(sizes) != 1 or local % self.parts:
      raise ValueError('PipelinedExchange needs equal shards that divide into %d equal parts (got shard sizes %s)'
                       % (self.parts, sorted(sizes)))
    self.part_size = local // self.parts
    # local row ranges of this rank's parts, and the global env index of every row of part p's gathered matrix
    self.part_bounds = [(p * self.part_size, (p + 1) * self.part_size) for p in range(self.parts)]
    self._buffers = {}
    self._pending_gather = [None] * self.parts
    self._pending_scatter = [None] * self.parts

  def global_rows(self, p):
    """Environment index (in the global batch) of every row of part p's gathered matrix / global action matrix."""
    rows = []
    for r in range(self.world):
      lo = shard_bounds(self.part_size * self.parts * self.world, self.world, r)[0] + p * self.part_size
      rows.extend(range(lo, lo + self.part_size))
    return np.asarray(rows)

  def _buffer(self, tag, shape, dtype):
    key = (tag, tuple(shape), dtype)
    b = self._buffers.get(key)
    if b is None:
      b = self._buffers[key] = self._torch.zeros(tuple(shape), dtype=dtype, device=self.device)
    return b

  def gather_async(self, p, local):
    """local: (part_size, n) observations of this rank's part p (consumed when the call returns: it is copied into the
    part's own send buffer, so the caller's tensor -- e.g. a HIP graph's output -- may be rewritten by the next step)."""
    tail = tuple(local.shape[1:])
    send = self._buffer(('obs_send', p), (self.part_size,) + tail, local.dtype)
    send.copy_(local)
    if self.world == 1:
      self._pending_gather[p] = (None, send)
      return
    recv = self._buffer(('obs_recv', p), (self.world * self.part_size,) + tail, local.dtype)
    self._pending_gather[p] = (self.dist.all_gather_into_tensor(recv, send, async_op=True), recv)

  def gather_wait(self, p):
    work, recv = self._pending_gather[p]
    self._pending_gather[p] = None
    if work is not None:
      work.wait()
    return recv

  def scatter_async(self, p, actions_global, nu, src=0):
    """actions_global: (world * part_size, nu) float32 on rank `src` (None elsewhere), rows as `global_rows(p)`."""
    torch = self._torch
    out = self._buffer(('act_recv', p), (self.part_size, nu), torch.float32)
    if self.world == 1:
      out.copy_(actions_global)
      self._pending_scatter[p] = (None, out)
      return
    chunks = None
    if self.rank == src:
      stage = self._buffer(('act_stage', p), (self.world * self.part_size, nu), torch.float32)
      stage.copy_(actions_global)
      chunks = list(stage.split(self.part_size))
    self._pending_scatter[p] = (self.dist.scatter(out, chunks, src=src, async_op=True), out)

  def scatter_wait(self, p):
    work, out = self._pending_scatter[p]
    self._pending_scatter[p] = None
    if work is not None:
      work.wait()
    return out
