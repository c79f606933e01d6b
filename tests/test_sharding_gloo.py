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


def _policy(obs, nu):
  """A deterministic 'agent': actions from the observations of the same environments (a_{t+1} = pi(o_t))."""
  import torch
  w = torch.linspace(-1, 1, obs.shape[1] * nu, dtype=torch.float64).reshape(obs.shape[1], nu)
  return torch.tanh(obs.double() @ w).float()


def _pipelined_worker(rank, world, port, B, T, parts, out_path):
  """PipelinedExchange: every rank steps its shard as `parts` part-batches; part p's all-gather / scatter are in flight
  while part p + 1 steps.  The policy lives on rank 0."""
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
    ex = sharding.PipelinedExchange(B, parts=parts, dist=dist)
    envs = []
    for e in range(ex.lo, ex.hi):
      p = EmuPhysics(m, 64)
      p.qpos[:] = m.qpos0
      p.qpos[3:] += np.random.RandomState(e).uniform(-.3, .3, 6)
      envs.append(p)

    def observe(part):
      lo, hi = ex.part_bounds[part]
      return torch.from_numpy(np.stack([np.r_[p.qpos, p.qvel] for p in envs[lo:hi]]))
    # t = 0: observations of the start states go out, the first actions come back
    for part in range(parts):
      ex.gather_async(part, observe(part))
    for part in range(parts):
      o = ex.gather_wait(part)
      ex.scatter_async(part, _policy(o, m.nu) if rank == 0 else None, m.nu)
    last = [None] * parts
    for t in range(T):
      for part in range(parts):
        a = ex.scatter_wait(part)
        lo, hi = ex.part_bounds[part]
        for p, act in zip(envs[lo:hi], a.numpy()):
          p.ctrl[:] = act
          p.step()
        ex.gather_async(part, observe(part))      # in flight while the next part steps
      for part in range(parts):
        o = ex.gather_wait(part)
        last[part] = o.clone()
        ex.scatter_async(part, _policy(o, m.nu) if rank == 0 else None, m.nu)
    for part in range(parts):
      ex.scatter_wait(part)
    if rank == 0:
      full = np.zeros((B, m.nq + m.nv))
      for part in range(parts):
        full[ex.global_rows(part)] = last[part].numpy()
      np.save(out_path, full)
    dist.barrier()
  finally:
    dist.destroy_process_group()


def test_pipelined_exchange_equals_the_serial_order(tmp_path):
  """world_size 2 x 2 part-batches over gloo, the exchange of one part overlapping the other part's steps: every
  environment's trajectory under an observation-driven policy equals the one-process serial loop bit for bit."""
  import sys
  import torch
  import torch.multiprocessing as mp
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from emu_lib import EmuPhysics
  from dm_control_amd import mjcf_compiler as mc
  from dm_control_amd.suite import common
  B, T, parts = 8, 5, 2
  ex = sharding.PipelinedExchange(B, parts=parts)      # (world 1: rows of part p are the envs [p B / P, (p + 1) B / P))
  assert list(ex.global_rows(1)) == [4, 5, 6, 7]
  with pytest.raises(ValueError):
    sharding.PipelinedExchange(7, parts=2)
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  out = str(tmp_path / 'pipelined.npy')
  mp.spawn(_pipelined_worker, args=(2, port, B, T, parts, out), nprocs=2, join=True)
  got = np.load(out)
  m = mc.compile_xml(common.read_model('cheetah.xml'))
  envs = []
  for e in range(B):
    p = EmuPhysics(m, 64)
    p.qpos[:] = m.qpos0
    p.qpos[3:] += np.random.RandomState(e).uniform(-.3, .3, 6)
    envs.append(p)
  # the serial loop: the policy sees part p's rows only (it is evaluated per part in the pipelined loop as well -- the
  # same function of the same rows)
  rows = [np.r_[0:2, 4:6], np.r_[2:4, 6:8]]      # global envs of part 0 / 1 with two ranks of four
  obs = np.stack([np.r_[p.qpos, p.qvel] for p in envs])
  for t in range(T):
    acts = np.zeros((B, m.nu), np.float32)
    for r in rows:
      acts[r] = _policy(torch.from_numpy(obs[r]), m.nu).numpy()
    for p, a in zip(envs, acts):
      p.ctrl[:] = a
      p.step()
    obs = np.stack([np.r_[p.qpos, p.qvel] for p in envs])
  np.testing.assert_array_equal(got, obs)
