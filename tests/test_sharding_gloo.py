"""N > 1 path on CPU: world_size-2 gloo processes exercise the env sharding and
the agent-interface collectives (action scatter, observation gather, max-time)."""
import os
import socket

import numpy as np
import pytest

from dm_control_amd import sharding


def test_shard_bounds_cover_batch():
  for B in (1, 7, 4096, 32768 + 3):
    for W in (1, 2, 3, 8):
      spans = [sharding.shard_bounds(B, W, r) for r in range(W)]
      assert spans[0][0] == 0 and spans[-1][1] == B
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    sharding.shard_bounds(8, 2, 2)


def _worker(rank, world, port, B, nu, nobs):
  import torch
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    sb = sharding.ShardedEnvBatch(B, dist)
    actions = torch.arange(B * nu, dtype=torch.float32).reshape(B, nu) if rank == 0 else None
    local = sb.scatter_actions(actions)
    want = torch.arange(B * nu, dtype=torch.float32).reshape(B, nu)[sb.lo:sb.hi]
    assert torch.equal(local, want), (rank, local, want)
    # "physics": obs_e = [sum(action_e), env index]
    idx = torch.arange(sb.lo, sb.hi, dtype=torch.float32)
    obs = torch.stack([local.sum(dim=1), idx] + [idx * 0] * (nobs - 2), dim=1)
    full = sb.gather(obs)
    assert full.shape == (B, nobs)
    assert torch.equal(full[:, 1], torch.arange(B, dtype=torch.float32))
    assert torch.allclose(full[:, 0], torch.arange(B * nu, dtype=torch.float32).reshape(B, nu).sum(dim=1))
    assert sb.max_over_ranks(1.0 + rank) == float(world)
    dist.barrier()
  finally:
    dist.destroy_process_group()


def test_two_rank_scatter_gather_gloo():
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_worker, args=(2, port, 11, 6, 4), nprocs=2, join=True)
