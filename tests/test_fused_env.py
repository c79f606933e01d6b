"""suite/fused_env.py: the host ports' task code compiled into ONE HIP kernel per task (trace -> expression DAG ->
generated C++), per-environment episode ends and restarts on the device.

CPU tier (oracle stand-in): for all 45 tasks the DAG traced from `get_observation / get_reward` evaluates to exactly what
the host port computes from the same state (the trace IS the port's single-environment code path); a task that branches
on a device value is refused; the generated source cross-compiles.  `-m gpu`: the generated kernels on the HIP batch --
observation / reward against the host port from the same device state, per-environment termination (lqr), time limits,
restarts from the pool in the same launch as the others step, eager == captured graph."""
import os

import numpy as np
import pytest

from dm_control_amd import suite
from dm_control_amd.suite import fused_env

ALL_TASKS = sorted((d, t) for d, t in suite.ALL_TASKS)


def _port_eval(env, B):
  p = env.physics
  p.data._invalidate()
  obs = np.concatenate([np.asarray(v, dtype=np.float64).reshape(B, -1) for v in env.task.get_observation(p).values()], axis=1)
  return obs, np.broadcast_to(np.asarray(env.task.get_reward(p), dtype=np.float64), (B,))


@pytest.mark.parametrize('domain,task', ALL_TASKS)
def test_traced_dag_equals_the_host_port(oracle_backend, domain, task):
  B = 3
  env = suite.load(domain, task, task_kwargs=dict(random=4), physics_kwargs=dict(batch_size=B, precision=64))
  env.reset()
  rs = np.random.RandomState(1)
  for _ in range(3):
    env.step(rs.uniform(-1, 1, (B, env.physics.model.nu)))
  prog = fused_env.trace(env)
  p = env.physics
  attrs = fused_env.episode_attrs(p, B)
  vals = fused_env.evaluate(prog.graph, prog.obs_nodes + [prog.reward_node, prog.term_node],
                            lambda name, row: np.asarray(p.batch.get(name))[:, row], lambda name, j: attrs[name][0][:, j])
  obs = np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), (B,)) for v in vals[:-2]], axis=1)
  want_obs, want_rew = _port_eval(env, B)
  np.testing.assert_allclose(obs, want_obs, rtol=0, atol=1e-12)
  np.testing.assert_allclose(np.broadcast_to(np.asarray(vals[-2], dtype=np.float64), (B,)), want_rew, rtol=0, atol=1e-12)
  assert prog.nobs == want_obs.shape[1] and list(prog.observation_layout) == list(env.task.get_observation(p))
  if domain == 'lqr':      # per-environment termination: the reference's test (suite/lqr.py:264) for each environment
    norm = np.linalg.norm(np.concatenate([p.batch.get('qpos'), p.batch.get('qvel')], axis=1), axis=1)
    np.testing.assert_array_equal(np.broadcast_to(vals[-1], (B,)), norm < 1e-6)
    assert prog.term_discount == 0.0
  else:
    assert prog.term_node.op == 'const' and prog.term_node.args[0] is False


def test_trace_refuses_host_decisions_and_folds_constants(oracle_backend):
  env = suite.load('cartpole', 'swingup', task_kwargs=dict(random=0), physics_kwargs=dict(batch_size=2, precision=64))
  env.reset()

  class Branching(type(env.task)):
    def get_reward(self, physics):
      if physics.cart_position() > 0:      # a python branch on a per-environment value
        return 1.0
      return 0.0
  env._task = Branching(swing_up=True, sparse=False, random=0)
  with pytest.raises(TypeError, match='device value'):
    fused_env.trace(env)
  g = fused_env.Graph()
  assert g.binary('add', 2.0, 3.0).args == (5.0,) and g.unary('sqrt', 4.0).args == (2.0,)
  x = g.load('qpos', 0)
  assert g.binary('mul', x, 1.0) is x and g.binary('pow', x, 2.0) is g.binary('mul', x, x) and g.where(True, x, 0.0) is x
  assert g.const(True).kind == 'b' and g.const(1.0).kind == 'f' and g.const(True) is not g.const(1.0)


@pytest.mark.parametrize('domain,task,precision', [('cartpole', 'swingup', 32), ('humanoid_CMU', 'run', 32), ('lqr', 'lqr_6_2', 64)])
def test_generated_source_cross_compiles(oracle_backend, tmp_path, monkeypatch, domain, task, precision):
  import subprocess
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  env = suite.load(domain, task, task_kwargs=dict(random=4), physics_kwargs=dict(batch_size=2, precision=64))
  env.reset()
  prog = fused_env.trace(env, precision=precision, title='%s.%s' % (domain, task))
  path = prog.build()
  syms = subprocess.run(['nm', '-D', path], capture_output=True, text=True).stdout
  assert 'fused_restart' in syms and 'fused_post' in syms and os.path.exists(prog.header_path)
  assert ('typedef float T' if precision == 32 else 'typedef double T') in prog.source
  assert prog.build() == path


def _make(domain, task, B, **kw):
  kw.setdefault('precision', 64)
  kw.setdefault('inline', False)      # (the 45-task sweeps use the stand-alone kernel: no 20 s kernel build per task)
  return fused_env.make(domain, task, B, seed=3, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize('domain,task', [('cartpole', 'swingup'), ('cheetah', 'run'), ('reacher', 'hard'), ('humanoid', 'stand'), ('lqr', 'lqr_2_1'),
                                         ('finger', 'turn_hard')])
def test_task_layer_inside_the_step_kernel_equals_the_separate_kernel(domain, task, tmp_path, monkeypatch):
  """`inline=True`: the generated function as the epilogue of a step kernel specialised for (model, task) -- the whole
  Environment.step in ONE launch -- against the same function as a kernel of its own behind the physics launch: identical
  observations, rewards, flags, counters and restarts over steps that cross episode ends."""
  import torch
  B = 40
  outs = []
  for inline in (True, False):
    # fp64: the kernel specialised for (model, task) and the library's kernel order some sums differently -- equal to
    # rounding, which is what is asserted (fp32 trajectories of the two drift apart at the first just-touching contact)
    env = _make(domain, task, B, precision=64, inline=inline, task_kwargs=dict(time_limit=5 * (0.01 if domain in ('cartpole', 'cheetah') else 0.02 if domain in ('reacher', 'finger') else 0.025 if domain == 'humanoid' else 0.03)))
    assert env.inline == inline and env.step_limit <= 6
    assert (env.host_physics.batch.info()['static_id'] == 1000) == inline
    rs = np.random.RandomState(9)
    log = []
    for k in range(13):
      obs, rew, done = env.step(torch.as_tensor(rs.uniform(-1, 1, (B, env.model.nu)), device='cuda'))
      torch.cuda.synchronize()
      log.append([x.cpu().numpy().copy() for x in (obs, rew, done, env.first, env.discount, env.terminated, env.steps, env.episode)])
    outs.append(log)
    assert not env.warnings().any()
    env.close()
  assert np.stack([l[2] for l in outs[0]]).any()      # (episodes did end)
  for a, b in zip(*outs):
    for k, (x, y) in enumerate(zip(a, b)):
      if k < 2:
        np.testing.assert_allclose(x, y, rtol=0, atol=1e-9)
      else:
        np.testing.assert_array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize('domain,task', ALL_TASKS)
def test_generated_kernels_equal_the_host_port_on_the_device(domain, task):
  """Observation and reward of the generated kernel against the host port evaluated (through the facade) on the same device
  state, over steps that include the first step of the episodes (mj_forward under the launch override) and stepping."""
  import torch
  B = 32
  env = _make(domain, task, B)
  rs = np.random.RandomState(5)
  for k in range(4):
    a = torch.as_tensor(rs.uniform(-1, 1, (B, env.model.nu)), device='cuda')
    obs, rew, done = env.step(a)
    torch.cuda.synchronize()
    assert bool(env.first.bool().all()) == (k == 0)
    for name, live in env._attr_live.items():      # the port reads the episode's attributes (targets ...) off the physics
      host = getattr(env.host_physics, name)
      setattr(env.host_physics, name, live.cpu().numpy().astype(np.float64).reshape(host.shape))
    want_obs, want_rew = _port_eval(env.host_env, B)
    np.testing.assert_allclose(obs.cpu().numpy(), want_obs, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rew.cpu().numpy(), want_rew if k else 0 * want_rew, rtol=1e-9, atol=1e-9)
    assert not bool(done.any())
  assert not env.warnings().any()
  env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('capture', [False, True])
def test_lqr_ends_each_environment_at_its_own_step(capture):
  """suite/lqr.py:264 + rl/control.py:114-127 per environment: the state norm of SOME environments is put below the
  tolerance -- exactly those report done / terminated with discount 0 in that step and start a new episode (first, reward 0,
  step count 0, a pool state) in the next one, in the same launch in which the others keep stepping."""
  import torch
  B = 64
  env = _make('lqr', 'lqr_2_1', B, capture=capture)
  z = torch.zeros((B, env.model.nu), dtype=torch.float64, device='cuda')
  env.step(z); env.step(z)
  idx = [3, 17, 40]
  env._tensors['qpos'][:, idx] = 1e-9
  env._tensors['qvel'][:, idx] = 0.0
  q_before = env._tensors['qpos'].clone()
  obs, rew, done = env.step(z)
  torch.cuda.synchronize()
  want = np.zeros(B, bool); want[idx] = True
  np.testing.assert_array_equal(done.cpu().numpy(), want)
  np.testing.assert_array_equal(env.terminated.bool().cpu().numpy(), want)
  np.testing.assert_array_equal(env.discount.cpu().numpy(), np.where(want, 0.0, 1.0))
  np.testing.assert_array_equal(env.steps.cpu().numpy(), np.where(want, 0, 2))      # (two steps after the first; the ended episodes are re-armed)
  obs2, rew2, done2 = env.step(z)
  torch.cuda.synchronize()
  np.testing.assert_array_equal(env.first.bool().cpu().numpy(), want)
  assert not bool(done2.any())
  np.testing.assert_array_equal(env.steps.cpu().numpy(), np.where(want, 0, 3))
  np.testing.assert_array_equal(rew2.cpu().numpy()[idx], 0.0)
  assert (rew2.cpu().numpy()[~want] != 0).all()
  # the restarted environments hold one of their pool states; the others moved on from their own state
  q = env._tensors['qpos'].cpu().numpy()
  pool = env._pool['qpos'].cpu().numpy()
  for e in idx:
    assert any(np.array_equal(q[:, e], pool[r, :, e]) for r in range(env.rounds)), e
  assert np.abs(q[:, ~want] - q_before.cpu().numpy()[:, ~want]).max() > 0
  env.close()


@pytest.mark.gpu
def test_time_limit_per_environment_and_captured_equals_eager():
  """A 5-step time limit: every environment reports done at its fifth step and first at the next; an environment made
  to restart early runs out of phase with the rest from then on.  The captured HIP graph replays the eager loop bit for bit
  (same pool, same episode counters)."""
  import torch
  B = 48
  outs = []
  for capture in (False, True):
    env = _make('reacher', 'hard', B, capture=capture, task_kwargs=dict(time_limit=0.1))      # control step 0.02 s: 5 steps
    assert env.step_limit == 5
    rs = np.random.RandomState(0)
    log = []
    for k in range(14):
      if k == 3:
        m = torch.zeros(B, dtype=torch.bool, device='cuda'); m[7] = True
        env.restart(m)      # environment 7 restarts at step 3
      obs, rew, done = env.step(torch.as_tensor(rs.uniform(-1, 1, (B, env.model.nu)), device='cuda'))
      torch.cuda.synchronize()
      log.append((obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), env.first.cpu().numpy().copy(), env.steps.cpu().numpy().copy()))
    outs.append(log)
    env.close()
  for a, b in zip(*outs):
    for x, y in zip(a, b):
      np.testing.assert_array_equal(x, y)
  done = np.stack([l[2] for l in outs[0]])      # (step, env)
  first = np.stack([l[3] for l in outs[0]]).astype(bool)
  others = np.arange(B) != 7
  # step 0 is the first of every episode; the fifth step after it ends it; the next one is a first again
  assert first[0].all() and done[5, others].all() and first[6, others].all() and done[11, others].all() and first[12, others].all()
  assert not done[[0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 12, 13]][:, others].any()
  assert first[3, 7] and done[8, 7] and first[9, 7] and not done[5, 7]
  # per-episode attributes travel with the restart: the reacher's target is one of the pool's for that environment
  assert np.isfinite(outs[0][-1][0]).all()
