"""suite/device_env.py: every suite task's OWN host task code (`get_observation / get_reward` of dm_control_amd/suite/*)
evaluated on torch tensors through numpy's dispatch protocols (45 / 45 tasks device-resident).

CPU tier: torch CPU tensors next to the oracle stand-in -- what is tested is the dispatch layer (`TArr`) against numpy
itself, task by task: the same state gives the same observation and reward to 1e-12, and no value is read back to the
host on the way (`TArr.host_reads` stays put).  `-m gpu`: the tensors are the HIP batch's bound fields, the control step
is a captured HIP graph, and the observations / rewards equal the host `Environment`'s from the same state."""
import numpy as np
import pytest

from dm_control_amd import suite
from dm_control_amd.suite import device_env

ALL_TASKS = sorted((d, t) for d, t in suite.ALL_TASKS)


def _host_eval(env):
  p = env.host_physics
  p.data._invalidate()
  obs = env.task.get_observation(p)
  flat = np.concatenate([np.asarray(v, dtype=np.float64).reshape(env.B, -1) for v in obs.values()], axis=1)
  return flat, np.broadcast_to(np.asarray(env.task.get_reward(p), dtype=np.float64), (env.B,))


@pytest.mark.parametrize('domain,task', ALL_TASKS)
def test_task_code_on_tensors_equals_task_code_on_numpy(oracle_backend, domain, task):
  import torch
  B = 3
  env = device_env.make(domain, task, B, precision=64, _device='cpu', capture=False, seed=4)
  rs = np.random.RandomState(2)
  reads = device_env.TArr.host_reads
  for k in range(3):
    a = torch.as_tensor(rs.uniform(-1, 1, (B, env.model.nu)))
    obs, rew, done = env.step(a)
    assert device_env.TArr.host_reads == reads, 'the task layer read a device value back'
    want_obs, want_rew = _host_eval(env)
    np.testing.assert_allclose(obs.numpy(), want_obs, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(rew.numpy(), want_rew, rtol=1e-12, atol=1e-12)
    reads = device_env.TArr.host_reads
    assert obs.shape == (B, sum(int(np.prod(s)) for s in env.observation_layout.values()))
  env.close()


def test_restart_refreshes_the_per_episode_device_copies_in_place(oracle_backend):
  """Targets the task hangs on the physics at episode start (reacher: `target_xy`) become device tensors that a restart
  rewrites IN PLACE: a captured graph keeps reading the same memory."""
  import torch
  env = device_env.make('reacher', 'easy', 4, precision=64, _device='cpu', capture=False, seed=1)
  env.step(torch.zeros(4, env.model.nu, dtype=torch.float64))
  t1 = env.view.target_xy.t
  first = t1.clone()
  env.reset()
  t2 = env.view.target_xy.t
  assert t2.data_ptr() == t1.data_ptr() and not torch.equal(first, t2)
  np.testing.assert_allclose(t2.numpy(), env.host_physics.target_xy)
  env.close()


def test_single_environment_targets_are_refreshed_in_place_too(oracle_backend):
  """with B = 1 the ports drop the batch axis (`xy[0] if B == 1 else xy`): `target_xy` is a (2,) array, which
  must still become a device tensor rewritten in place by a restart -- as a host constant it would be baked into the
  captured graph at the first episode's value.  Two environments of different precision keep their own float type."""
  import torch
  env = device_env.make('reacher', 'easy', 1, precision=64, _device='cpu', capture=False, seed=1)
  other = device_env.make('reacher', 'easy', 2, precision=32, _device='cpu', capture=False, seed=1)
  env.step(torch.zeros(1, env.model.nu, dtype=torch.float64))
  t1 = env.view.target_xy.t
  first = t1.clone()
  env.reset()
  t2 = env.view.target_xy.t
  assert t2.data_ptr() == t1.data_ptr() and not torch.equal(first, t2) and tuple(t2.shape) == (2,)
  np.testing.assert_allclose(t2.numpy(), env.host_physics.target_xy)
  obs, rew, _ = env.step(torch.zeros(1, env.model.nu, dtype=torch.float64))
  assert obs.dtype == torch.float64 and rew.dtype == torch.float64      # (the fp32 environment built after it did not change that)
  o32, r32, _ = other.step(torch.zeros(2, other.model.nu, dtype=torch.float32))
  assert o32.dtype == torch.float32 and env._consts is not other._consts
  env.close(); other.close()


@pytest.mark.gpu
@pytest.mark.parametrize('domain,task', ALL_TASKS)
def test_device_env_matches_the_host_task_on_the_device(domain, task):
  import torch
  B = 8
  env = device_env.make(domain, task, B, precision=64, seed=3, capture=True)
  g = torch.Generator(device='cuda').manual_seed(1)
  for k in range(4):
    a = torch.rand((B, env.model.nu), device='cuda', generator=g, dtype=torch.float64) * 2 - 1
    obs, rew, done = env.step(a)
    torch.cuda.synchronize()
    want_obs, want_rew = _host_eval(env)
    np.testing.assert_allclose(obs.cpu().numpy(), want_obs, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rew.cpu().numpy(), want_rew, rtol=1e-9, atol=1e-9)
  assert not env.warnings()[:, :8].any()
  env.close()


@pytest.mark.gpu
def test_device_env_restarts_at_the_time_limit_and_replays_its_graph():
  import torch
  env = device_env.make('pendulum', 'swingup', 16, precision=32, seed=0, capture=True)
  a = torch.zeros((16, env.model.nu), device='cuda')
  n = int(np.ceil(env.step_limit))
  q0 = env._tensors['qpos'].clone()
  for k in range(n):
    obs, rew, done = env.step(a)
    assert bool(done.all()) == (k == n - 1)
  assert env.steps == 0 and not torch.equal(q0, env._tensors['qpos'])
  obs2, _, _ = env.step(a)      # the captured graph keeps working after the restart
  assert torch.isfinite(obs2).all()
  env.close()
