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
    # per-step calls reuse their buffers (no allocation after the first exchange)
    nbuf = len(sb._buffers)
    ptrs = (local.data_ptr(), full.data_ptr())
    local2 = sb.scatter_actions(actions)
    full2 = sb.gather(obs)
    assert len(sb._buffers) == nbuf and (local2.data_ptr(), full2.data_ptr()) == ptrs
    assert torch.equal(local2, want) and torch.equal(full2[:, 1], torch.arange(B, dtype=torch.float32))
    dist.barrier()
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('B', [11, 12])       # ragged and even shardings (the even one works on views, no staging)
def test_two_rank_scatter_gather_gloo(B):
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_worker, args=(2, port, B, 6, 4), nprocs=2, join=True)


def _step_worker(rank, world, port, B, T, out_path):
  """Sharding and stepping together: each rank owns a contiguous env range and steps it with the host build of the
  kernel (tests/emu, one env at a time); actions come from rank 0, observations go back to every rank."""
  import sys
  import torch
  import torch.distributed as dist
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from emu_lib import EmuPhysics
  from dm_control_amd import mjcf_compiler as mc
  from dm_control_amd.suite import common
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    m = mc.compile_xml(common.read_model('cheetah.xml'))
    sb = sharding.ShardedEnvBatch(B, dist)
    envs = []
    for e in range(sb.lo, sb.hi):
      p = EmuPhysics(m, 64)
      p.qpos[:] = m.qpos0
      p.qpos[3:] += np.random.RandomState(e).uniform(-.3, .3, 6)
      envs.append(p)
    rs = np.random.RandomState(0)
    for t in range(T):
      acts = torch.from_numpy(rs.uniform(-1, 1, (B, m.nu)).astype(np.float32)) if rank == 0 else None
      local = sb.scatter_actions(acts)
      assert tuple(local.shape) == (sb.local_batch, m.nu)
      for p, a in zip(envs, local.numpy()):
        p.ctrl[:] = a
        p.step()
      obs = torch.from_numpy(np.stack([np.r_[p.qpos, p.qvel] for p in envs]))
      full = sb.gather(obs)
      assert tuple(full.shape) == (B, m.nq + m.nv)
    if rank == 0:
      np.save(out_path, full.numpy())
    dist.barrier()
  finally:
    dist.destroy_process_group()


def test_sharded_batch_steps_like_one_batch(tmp_path):
  """world_size 2 over gloo: ShardedEnvBatch + physics stepping == the same environments stepped in one process."""
  import sys
  import torch.multiprocessing as mp
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from emu_lib import EmuPhysics
  from dm_control_amd import mjcf_compiler as mc
  from dm_control_amd.suite import common
  B, T = 5, 6
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  out = str(tmp_path / 'sharded.npy')
  mp.spawn(_step_worker, args=(2, port, B, T, out), nprocs=2, join=True)
  got = np.load(out)
  m = mc.compile_xml(common.read_model('cheetah.xml'))
  rs = np.random.RandomState(0)
  envs = []
  for e in range(B):
    p = EmuPhysics(m, 64)
    p.qpos[:] = m.qpos0
    p.qpos[3:] += np.random.RandomState(e).uniform(-.3, .3, 6)
    envs.append(p)
  for t in range(T):
    acts = rs.uniform(-1, 1, (B, m.nu)).astype(np.float32)
    for p, a in zip(envs, acts):
      p.ctrl[:] = a
      p.step()
  want = np.stack([np.r_[p.qpos, p.qvel] for p in envs])
  np.testing.assert_array_equal(got, want)
