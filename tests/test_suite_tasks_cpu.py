"""Host stack without a GPU: `suite.load`, the Physics facade, named indexing, the Environment loop and every
suite task run on the oracle-backed stand-in batch (tests/oracle_backend.py).  Mirrors the reference's
suite-level properties (dm_control/suite/suite_test.py: spec conformance :149, determinism :170, finite
observations :81, reward range :94).  The `-m gpu` suite runs the same checks on the HIP backend."""
import numpy as np
import pytest

from dm_control_amd import suite

_SLOW = {'humanoid_CMU'}          # a 62-dof oracle step is ~2 ms: fewer env-steps there


@pytest.mark.parametrize('domain,task', suite.ALL_TASKS)
def test_task_runs_and_respects_its_specs(oracle_backend, domain, task):
  env = suite.load(domain, task, task_kwargs=dict(random=0))
  aspec, ospec = env.action_spec(), env.observation_spec()
  rs = np.random.RandomState(1)
  ts = env.reset()
  assert ts.first() and ts.reward is None
  assert set(ts.observation) == set(ospec)
  for _ in range(3 if domain in _SLOW else 8):
    lo, hi = np.maximum(aspec.minimum, -1), np.minimum(aspec.maximum, 1)      # lqr's action spec is unbounded
    ts = env.step(rs.uniform(lo, hi, aspec.shape))
    for k, v in ts.observation.items():
      v = np.asarray(v)
      assert v.shape == ospec[k].shape and np.all(np.isfinite(v)), (k, v.shape, ospec[k].shape)
    assert float(ts.reward) <= 1.0 and (domain == 'lqr' or float(ts.reward) >= 0.0)   # lqr: 1 - quadratic cost
  env.physics.free()


@pytest.mark.parametrize('domain,task', [('cheetah', 'run'), ('humanoid', 'walk'), ('quadruped', 'run'), ('stacker', 'stack_2'),
                                         ('manipulator', 'insert_peg'), ('finger', 'turn_hard'), ('fish', 'swim')])
def test_same_seed_same_trajectory_and_batches_agree_with_single_envs(oracle_backend, domain, task):
  def rollout(batch):
    env = suite.load(domain, task, task_kwargs=dict(random=42), physics_kwargs=dict(batch_size=batch))
    aspec = env.action_spec()
    rs = np.random.RandomState(7)
    ts = env.reset()
    out = [np.concatenate([np.reshape(v, (batch, -1)) for v in ts.observation.values()], axis=1)]
    for _ in range(5):
      ts = env.step(rs.uniform(aspec.minimum, aspec.maximum, aspec.shape))
      out.append(np.concatenate([np.reshape(v, (batch, -1)) for v in ts.observation.values()] + [np.reshape(ts.reward, (batch, 1))], axis=1))
    env.physics.free()
    return out
  a, b = rollout(1), rollout(1)
  for x, y in zip(a, b):
    np.testing.assert_array_equal(x, y)
  # a batch of 3 produces well-formed (3, ...) observations and rewards in [0, 1]
  c = rollout(3)
  assert c[0].shape[0] == 3 and c[0].shape[1] == a[0].shape[1]
  assert all(np.isfinite(x).all() for x in c)
  assert ((c[-1][:, -1] >= 0) & (c[-1][:, -1] <= 1)).all()


def test_humanoid_cmu_observation_formulas(oracle_backend):
  env = suite.load('humanoid_CMU', 'walk', task_kwargs=dict(random=3))
  ts = env.reset()
  p = env.physics
  m = p.model
  bid = lambda n: m.name2id(n, 'body')
  xpos = np.asarray(p.data.xpos).reshape(-1, 3)
  xmat = np.asarray(p.data.xmat).reshape(-1, 3, 3)
  R, t = xmat[bid('thorax')], xpos[bid('thorax')]
  ext = np.concatenate([(xpos[bid(s + l)] - t) @ R for s in ('l', 'r') for l in ('hand', 'foot')])
  np.testing.assert_allclose(ts.observation['extremities'], ext, atol=1e-12)
  np.testing.assert_allclose(ts.observation['head_height'], xpos[bid('head'), 2], atol=1e-12)
  np.testing.assert_allclose(ts.observation['torso_vertical'], R[2], atol=1e-12)
  np.testing.assert_allclose(p.thorax_upright(), R[2, 1], atol=1e-12)
  np.testing.assert_allclose(ts.observation['joint_angles'], np.asarray(p.data.qpos)[7:], atol=0)
  assert np.asarray(p.data.ncon) == 0                      # rejection-sampled start: nothing touches
  env.physics.free()


def test_reload_from_xml_string_keeps_the_object_and_its_settings(oracle_backend):
  # composer/environment.py:377-383 recompiles and reloads at episode boundaries
  from dm_control_amd.suite import cheetah, walker
  p = cheetah.Physics.from_xml_string(*cheetah.get_model_and_assets(), batch_size=2, nconmax=20)
  p.legacy_step = False
  p.set_control(np.ones((2, 6)))
  p.step(5)
  assert p.model.nq == 9 and np.asarray(p.data.time).max() > 0
  p.reload_from_xml_string(walker.get_model_and_assets()[0])
  assert type(p) is cheetah.Physics and p.batch_size == 2 and p.legacy_step is False
  assert p.batch.nconmax == 20                        # caps travel with the object
  assert p.model.nq == 9 and p.model.nu == 6 and 'right_hip' in p.model.names['joint']
  assert np.asarray(p.data.time).max() == 0           # fresh state, named indexing rebuilt for the new model
  assert np.asarray(p.named.data.qpos['right_hip']).shape == (2, 1)
  p.step(3)
  with pytest.raises(NotImplementedError):
    p.render()
  p.free()


def test_gather_table_reproduces_task_observations(oracle_backend):
  """named-field observations resolved once into row indices; evaluated for a batch with one
  gather per field.  The cheetah and humanoid observations that are plain field slices come out identical."""
  from dm_control_amd.observation import GatherTable
  env = suite.load('cheetah', 'run', task_kwargs=dict(random=1), physics_kwargs=dict(batch_size=3))
  ts = env.reset()
  m = env.physics.model
  joints = [n for n in m.names['joint'] if n != 'rootx']
  table = GatherTable(m, [('qpos', joints), ('qvel', None), ('sensordata', ['torso_subtreelinvel']), ('xpos', ['torso'], 'z')])
  env.physics.data._upload()
  got = table.gather(env.physics)
  assert got.shape == (3, 8 + 9 + 3 + 1)
  np.testing.assert_array_equal(got[:, :8], ts.observation['position'])
  np.testing.assert_array_equal(got[:, 8:17], ts.observation['velocity'])
  np.testing.assert_array_equal(got[:, 17], env.physics.speed())
  np.testing.assert_array_equal(got[:, 20], np.asarray(env.physics.named.data.xpos['torso', 'z']))
  assert list(table.rows['xpos']) == [3 * m.name2id('torso', 'body') + 2]
  env.physics.free()
  env = suite.load('humanoid', 'stand', task_kwargs=dict(random=1))
  ts = env.reset()
  m = env.physics.model
  table = GatherTable(m, [('qpos', [n for n in m.names['joint'] if n != 'root']), ('xpos', ['head'], 'z'),
                          ('xmat', ['torso'], ['zx', 'zy', 'zz']), ('sensordata', ['torso_subtreelinvel']), ('qvel', None)])
  got = table.gather(env.physics)
  want = np.concatenate([ts.observation['joint_angles'], [ts.observation['head_height']], ts.observation['torso_vertical'],
                         ts.observation['com_velocity'], ts.observation['velocity']])
  np.testing.assert_array_equal(got, want)
  with pytest.raises(KeyError):
    GatherTable(m, [('qpos', ['no_such_joint'])])
  with pytest.raises(ValueError):
    GatherTable(m, [('efc_J', None)])
  env.physics.free()


# ---- suite/utils/randomizers_test.py mirrored on the oracle-backed facade -----------------------------------
def _physics(xml, **kw):
  from dm_control_amd import physics as physics_lib
  return physics_lib.Physics.from_xml_string(xml, **kw)


def test_randomizer_single_joint_of_each_type(oracle_backend):
  from dm_control_amd.suite import randomizers
  p = _physics("""<mujoco><default><joint range="0 90" armature="1"/></default><worldbody>
      <body><geom type="box" size="1 1 1"/><joint name="free" type="free"/></body>
      <body><geom type="box" size="1 1 1"/><joint name="limited_hinge" type="hinge" limited="true"/>
        <joint name="slide" type="slide" limited="false"/><joint name="limited_slide" type="slide" limited="true"/>
        <joint name="hinge" type="hinge" limited="false"/></body>
      <body><geom type="box" size="1 1 1"/><joint name="ball" type="ball" limited="false"/></body>
    </worldbody></mujoco>""")
  randomizers.randomize_limited_and_rotational_joints(p, np.random.RandomState(100))
  q = p.named.data.qpos
  assert q['hinge'] != 0 and q['limited_hinge'] != 0 and q['limited_slide'] != 0
  assert np.sum(q['ball']) != 0 and np.sum(q['free'][3:]) != 0
  np.testing.assert_allclose(np.linalg.norm(q['ball']), 1, atol=1e-12)
  np.testing.assert_allclose(np.linalg.norm(q['free'][3:]), 1, atol=1e-12)
  # the unlimited slide and the translation of the free joint are left alone
  assert q['slide'] == 0 and np.sum(q['free'][:3]) == 0
  p.free()


def test_randomizer_ranges_and_distinct_draws(oracle_backend):
  from dm_control_amd.suite import randomizers
  rand = np.random.RandomState(100)
  p = _physics("""<mujoco><worldbody><body><geom type="box" size="1 1 1"/>
      <joint name="hinge_1" type="hinge"/><joint name="hinge_2" type="hinge"/><joint name="hinge_3" type="hinge"/></body>
    </worldbody></mujoco>""")
  for _ in range(10):
    randomizers.randomize_limited_and_rotational_joints(p, rand)
    a, b, c = (float(p.named.data.qpos[n][0]) for n in ('hinge_1', 'hinge_2', 'hinge_3'))
    assert len({a, b, c}) == 3 and all(-np.pi <= v <= np.pi and v != 0 for v in (a, b, c))
  p.free()
  p = _physics("""<mujoco><default><joint limited="true"/></default><worldbody><body><geom type="box" size="1 1 1"/>
      <joint name="hinge" type="hinge" range="0 10"/><joint name="slide" type="slide" range="30 50"/></body>
    </worldbody></mujoco>""")
  for _ in range(10):
    randomizers.randomize_limited_and_rotational_joints(p, rand)
    assert 0 <= p.named.data.qpos['hinge'][0] <= np.deg2rad(10)
    assert 30 <= p.named.data.qpos['slide'][0] <= 50
  p.free()
  # batch: only the masked environments are re-drawn
  p = _physics("""<mujoco><worldbody><body><geom type="box" size="1 1 1"/><joint name="h" type="hinge"/></body>
    </worldbody></mujoco>""", batch_size=4)
  randomizers.randomize_limited_and_rotational_joints(p, rand, env_mask=[True, False, True, False])
  q = np.asarray(p.data.qpos)[:, 0]
  assert q[0] != 0 and q[2] != 0 and q[1] == 0 and q[3] == 0
  p.free()


@pytest.mark.parametrize('domain,task', suite.ALL_TASKS)
def test_every_task_runs_as_a_batch(oracle_backend, domain, task):
  """batch_size = 2: observations gain a leading batch axis, rewards come back as (2,), both environments evolve
  (different random starts) and the first one equals... nothing in particular: only shapes and ranges are checked."""
  single = suite.load(domain, task, task_kwargs=dict(random=0))
  ospec = single.observation_spec()
  single.physics.free()
  env = suite.load(domain, task, task_kwargs=dict(random=0), physics_kwargs=dict(batch_size=2))
  aspec = env.action_spec()
  assert aspec.shape[0] == 2
  rs = np.random.RandomState(3)
  ts = env.reset()
  for _ in range(2 if domain == 'humanoid_CMU' else 4):
    lo, hi = np.maximum(aspec.minimum, -1), np.minimum(aspec.maximum, 1)
    ts = env.step(rs.uniform(lo, hi, aspec.shape))
    for k, v in ts.observation.items():
      v = np.asarray(v)
      assert v.shape == (2,) + ospec[k].shape and np.all(np.isfinite(v)), (k, v.shape, ospec[k].shape)
    r = np.asarray(ts.reward)
    assert r.shape == (2,) and np.all(r <= 1.0) and (domain == 'lqr' or np.all(r >= 0.0))
  env.physics.free()
