"""dm_control/mujoco/engine_test.py (the cases that do not render) against the Physics facade, run on the
oracle-backed stand-in batch (tests/oracle_backend.py) so that they are part of the `-m "not gpu"` tier."""
import copy
import os
import pickle

import numpy as np
import pytest

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control

CARTPOLE = """<mujoco model="cart-pole"><option timestep="0.01"/>
<worldbody>
  <geom name="floor" type="plane" pos="0 0 -.05" size="4 4 .2"/>
  <body name="cart" pos="0 0 1">
    <joint name="slider" type="slide" limited="true" axis="1 0 0" range="-1.8 1.8" damping="5e-4"/>
    <geom name="cart" type="box" size="0.2 0.15 0.1" mass="1"/>
    <body name="pole"><joint name="hinge_1" type="hinge" axis="0 1 0" damping="2e-6"/>
      <geom name="pole" type="capsule" fromto="0 0 0 0 0 1" size="0.045" mass=".1"/>
      <site name="tip" pos="0 0 1"/></body></body>
</worldbody>
<actuator><motor name="slide" joint="slider" gear="10" ctrllimited="true" ctrlrange="-1 1"/></actuator>
<sensor><accelerometer name="accelerometer" site="tip"/></sensor>
<keyframe><key name="hanging_down" qpos="0 1.57"/><key qpos="0.1 3.14" qvel="0.5 0"/></keyframe>
</mujoco>"""


@pytest.fixture
def physics(oracle_backend):
  p = physics_lib.Physics.from_xml_string(CARTPOLE)
  yield p
  p.free()


def test_named_views(physics):
  assert physics.control().shape == (1,) and physics.position().shape == (2,) and physics.velocity().shape == (2,)
  assert physics.activation().shape == (0,) and physics.state().shape == (4,)
  assert physics.time() == 0.0 and physics.timestep() == 0.01
  assert physics.named.data.xpos['cart'].shape == (3,)
  assert physics.named.data.xpos[['cart', 'pole']].shape == (2, 3)


def test_set_get_physics_state(physics):
  state = physics.get_state()
  physics.set_state(state)
  new = np.random.RandomState(0).random_sample(state.shape)
  physics.set_state(new)
  np.testing.assert_allclose(physics.get_state(), new)
  with pytest.raises(ValueError):
    physics.set_state(np.repeat(state, 2))


def test_reload_from_path_and_string(physics, tmp_path):
  path = os.path.join(str(tmp_path), 'cartpole.xml')
  with open(path, 'w') as f:
    f.write(CARTPOLE)
  physics.step(3)
  physics.reload_from_xml_path(path)
  assert physics.time() == 0 and physics.model.nq == 2
  physics.reload_from_xml_string(CARTPOLE)
  other = physics_lib.Physics.from_xml_path(path)
  np.testing.assert_array_equal(other.data.qpos, physics.data.qpos)
  other.free()


def test_reset_and_keyframes(physics):
  physics.step(5)
  physics.reset()
  assert physics.data.qpos[1] == 0 and physics.time() == 0
  physics.reset(keyframe_id=0)
  assert physics.data.qpos[1] == physics.model.key_qpos[0, 1] == 1.57
  physics.reset(keyframe_id=1)
  np.testing.assert_array_equal(physics.data.qvel, [0.5, 0])
  for bad in (-1, 2):
    with pytest.raises(ValueError):
      physics.reset(keyframe_id=bad)


@pytest.mark.parametrize('bad_value', [float('inf'), float('nan'), 1e15])
def test_bad_qpos_raises_physics_error(physics, bad_value):
  with pytest.raises(control.PhysicsError, match='mjWARN_BADQPOS'):
    with physics.reset_context():
      pass
    physics.data.qpos[0] = bad_value
    physics.step()
  physics.reset()
  physics.step()                     # a reset physics is valid again


def test_nan_control_and_suppression(physics):
  with physics.reset_context():
    pass
  physics.data.ctrl[0] = float('nan')
  with pytest.raises(control.PhysicsError, match='mjWARN_BADCTRL'):
    physics.step()
  physics.data.ctrl[0] = float('nan')
  with physics.suppress_physics_errors():
    physics.forward()                # warns instead of raising
  physics.data.ctrl[0] = float('nan')
  with pytest.raises(control.PhysicsError, match='mjWARN_BADCTRL'):
    physics.forward()


@pytest.mark.parametrize('clone', [copy.copy, copy.deepcopy, lambda p: pickle.loads(pickle.dumps(p))])
def test_copy_or_pickle_continues_identically(physics, clone):
  for _ in range(10):
    physics.set_control([0.3])
    physics.step()
  other = clone(physics)
  assert other is not physics and type(other) is type(physics)
  for _ in range(10):
    for p in (physics, other):
      p.set_control([-0.2])
      p.step()
  np.testing.assert_array_equal(other.get_state(), physics.get_state())
  np.testing.assert_array_equal(other.data.xpos, physics.data.xpos)
  assert other.time() == physics.time()
  # warnings seen before the copy do not raise in the copy (engine_test.py:578-584)
  other.free()


def test_forward_dynamics_after_reset_and_actuation_disabled_in_after_reset(physics):
  with physics.reset_context():
    pass
  # the accelerometer at the pole tip reads +g along z after a reset (forward was run)
  np.testing.assert_allclose(physics.named.data.sensordata['accelerometer'][2], 9.81, rtol=1e-6)
  physics.data.ctrl[0] = 1.0
  physics.after_reset()              # forward with actuation disabled
  assert physics.data.actuator_force[0] == 0.0
  physics.forward()
  assert physics.data.actuator_force[0] == 1.0


def test_action_spec_limits(oracle_backend):
  p = physics_lib.Physics.from_xml_string("""<mujoco><worldbody><body><geom type="sphere" size="0.1"/>
      <joint type="hinge" name="hinge"/></body></worldbody><actuator>
      <motor joint="hinge" ctrllimited="false"/><motor joint="hinge" ctrllimited="true" ctrlrange="-1 2"/>
    </actuator></mujoco>""")
  spec = physics_lib.action_spec(p)
  assert spec.dtype == float
  np.testing.assert_array_equal(spec.minimum, [-1e10, -1.0])       # mjMAXVAL
  np.testing.assert_array_equal(spec.maximum, [1e10, 2.0])
  p.free()


@pytest.mark.parametrize('integrator', ['Euler', 'RK4'])
def test_nstep_equals_repeated_single_steps(oracle_backend, integrator):
  xml = CARTPOLE.replace('timestep="0.01"', 'timestep="0.01" integrator="%s"' % integrator)
  a, b = physics_lib.Physics.from_xml_string(xml), physics_lib.Physics.from_xml_string(xml)
  for p in (a, b):
    with p.reset_context():
      p.data.qpos[1] = 0.3
    p.set_control([0.5])
  a.step(4)
  for _ in range(4):
    b.step()
  np.testing.assert_array_equal(a.get_state(), b.get_state())
  assert a.time() == b.time()
  a.free(); b.free()


# ---- dm_control/mujoco/index_test.py: named indexing semantics ------------------------------------------------
def test_named_indexing_keys_match_numeric_indexing(physics):
  physics.step(3)
  d, n = physics.data, physics.named.data
  xpos, xmat, sd = np.asarray(d.xpos), np.asarray(d.xmat), np.asarray(d.sensordata)
  np.testing.assert_array_equal(n.xpos['pole'], xpos[2])
  np.testing.assert_array_equal(n.xpos[['pole', 'cart']], xpos[[2, 1]])
  np.testing.assert_array_equal(n.sensordata['accelerometer'], sd[0:3])
  np.testing.assert_array_equal(n.xpos['pole', 'y'], xpos[2, 1])
  np.testing.assert_array_equal(n.xmat['cart', ['yy', 'zz']], xmat[1, [4, 8]])
  # two-dimensional named indexing follows numpy's rules (mujoco/index_test.py:132-134): two lists pair up element by
  # element, a column of names broadcasts against a row of column names
  np.testing.assert_array_equal(n.xpos[['pole', 'cart'], ['x', 'z']], xpos[[2, 1], [0, 2]])
  np.testing.assert_array_equal(n.xpos[[['pole'], ['cart']], ['x', 'z']], xpos[[[2], [1]], [0, 2]])
  np.testing.assert_array_equal(n.xpos[np.array(['pole', 'cart']).reshape(-1, 1), ['x', 'z']], xpos[[[2], [1]], [0, 2]])
  np.testing.assert_array_equal(n.xpos[:, 0], xpos[:, 0])                                          # plain slices pass through
  np.testing.assert_array_equal(n.qpos['slider'], np.asarray(d.qpos)[0:1])                         # ragged rows are slices
  np.testing.assert_array_equal(n.qvel[['slider', 'hinge_1']], np.asarray(d.qvel)[[0, 1]])
  m = physics.named.model
  np.testing.assert_array_equal(m.actuator_gear['slide'], physics.model.actuator_gear[0])
  np.testing.assert_array_equal(m.dof_armature[['slider', 'hinge_1']], physics.model.dof_armature[[0, 1]])
  with pytest.raises(KeyError):
    n.xpos['no_such_body']


@pytest.mark.parametrize('field,key', [('qpos', 'slider'), ('qvel', ['slider', 'hinge_1']), ('ctrl', 'slide')])
def test_named_assignment_reaches_the_state(physics, field, key):
  indexer = getattr(physics.named.data, field)
  shape = np.shape(indexer[key])
  new = 0.25 + np.arange(int(np.prod(shape))).reshape(shape) if shape else 0.25
  indexer[key] = new
  np.testing.assert_array_equal(indexer[key], new)
  physics.forward()                          # uploaded with the next pipeline call and still there afterwards
  np.testing.assert_array_equal(getattr(physics.named.data, field)[key], new)
  with pytest.raises(AttributeError):
    physics.data.xpos = 0                    # derived arrays are read-only


# ---- round-2 advisor findings -------------------------------------------------------------------------
def test_model_arrays_the_device_cannot_follow_are_read_only(oracle_backend):
  """Writes to model arrays outside the pushed set used to be silently ignored by the device (body_mass *= 100 gave a
  bit-identical trajectory): they now fail loudly; the pushed set stays writable."""
  from dm_control_amd import suite
  env = suite.load('cheetah', 'run', task_kwargs=dict(random=0))
  p = env.physics
  with pytest.raises(ValueError):
    p.model.body_mass[1] = 100.0
  with pytest.raises(ValueError):
    p.named.model.geom_friction['torso', 0] = 0.0
  p.model.dof_damping[3] = 2.5                     # in the pushed set
  p.forward()


def test_get_set_state_with_signature(oracle_backend):
  """engine.py:235-285 with `sig`: mj_getState / mj_setState component order (mjtState bits)."""
  from dm_control_amd import suite
  env = suite.load('cheetah', 'run', task_kwargs=dict(random=1))
  p = env.physics
  env.reset()
  env.step(np.full(6, 0.3))
  m = p.model
  TIME, QPOS, QVEL, ACT, WARM, CTRL, QFRC, XFRC = (1 << k for k in range(8))
  s = p.get_state(QPOS | QVEL)
  np.testing.assert_array_equal(s, np.r_[p.data.qpos, p.data.qvel])
  np.testing.assert_array_equal(s, p.get_state())                      # no activations: same as the legacy form
  full = p.get_state(TIME | QPOS | QVEL | ACT | WARM | CTRL | QFRC | XFRC)
  assert full.shape == (1 + m.nq + m.nv + 0 + m.nv + m.nu + m.nv + 6 * m.nbody,)
  assert full[0] == p.data.time and np.array_equal(full[1 + m.nq + 2 * m.nv:][:m.nu], p.data.ctrl)
  q = p.copy()
  for _ in range(5):
    p.step()
  p.set_state(full, TIME | QPOS | QVEL | ACT | WARM | CTRL | QFRC | XFRC)      # (no forward: it would overwrite the warm start)
  for _ in range(5):
    p.step(); q.step()
  np.testing.assert_array_equal(p.data.qpos, q.data.qpos)             # state complete: continues identically
  with pytest.raises(ValueError):
    p.set_state(full[:-1], TIME | QPOS | QVEL | ACT | WARM | CTRL | QFRC | XFRC)
  with pytest.raises(ValueError):
    p.get_state(0)


def test_copy_keeps_subclass_episode_state_and_owns_its_model(oracle_backend):
  """copy() / pickle of a suite Physics subclass keep its per-episode attributes (reacher target) and do not share the
  mutable model unless share_model=True (engine.py:287-304)."""
  import pickle
  from dm_control_amd import suite
  env = suite.load('reacher', 'easy', task_kwargs=dict(random=3))
  env.reset()
  p = env.physics
  want = np.array(p.finger_to_target())
  c = p.copy()
  np.testing.assert_array_equal(np.array(c.finger_to_target()), want)
  u = pickle.loads(pickle.dumps(p))
  np.testing.assert_array_equal(np.array(u.finger_to_target()), want)
  assert c.model is not p.model and p.copy(share_model=True).model is p.model
  c.model.dof_damping[0] = 9.0
  assert p.model.dof_damping[0] != 9.0


def test_facade_fetches_the_fields_a_loop_reads_in_one_round_trip(oracle_backend):
  """A host loop reads the same few mjData fields after every step: from the second iteration on the facade fetches
  them together (BatchedPhysics.get_many: one device-to-host copy and one wait) instead of field by field, and the values
  are those of single reads."""
  import numpy as np
  from dm_control_amd import physics as physics_lib
  from dm_control_amd.suite import common
  p = physics_lib.Physics.from_xml_string(common.read_model('cheetah.xml'), batch_size=3)
  q = physics_lib.Physics.from_xml_string(common.read_model('cheetah.xml'), batch_size=3)
  rs = np.random.RandomState(0)
  for t in range(4):
    c = rs.uniform(-1, 1, (3, 6))
    p.set_control(c); q.set_control(c)
    calls0 = getattr(p.batch, 'get_many_calls', 0)
    p.step(); q.step()
    a = (np.array(p.data.qpos), np.array(p.data.qvel), np.array(p.data.sensordata))
    calls = getattr(p.batch, 'get_many_calls', 0) - calls0
    # from the second pass on: ONE round trip per step -- the warning counters step() checks and the three reads together
    assert calls <= 2 if t == 0 else calls == 1, (t, calls)
    np.testing.assert_array_equal(a[1], q.batch.get('qvel'))
    np.testing.assert_array_equal(a[2], q.batch.get('sensordata'))
    np.testing.assert_array_equal(a[0], q.batch.get('qpos'))
  p.free(); q.free()


# ---- collision filter bits rewritten at run time (composer/initializers/prop_initializer.py:138-160) ---------------------
_FILTER_XML = """<mujoco><option timestep='0.002'/><worldbody>
<geom name='floor' type='plane' size='2 2 .1'/>
<body name='a' pos='0 0 .1'><freejoint/><geom name='ga' type='sphere' size='.1'/></body>
<body name='b' pos='.5 0 .1'><freejoint/><geom name='gb' type='sphere' size='.1'/></body>
<body name='c' pos='.5 0 .32'><freejoint/><geom name='gc' type='box' size='.1 .1 .1'/></body>
</worldbody></mujoco>"""


def check_collision_filter_edits(make, atol=0.0):
  """`make(xml)` -> Physics.  Switching a geom's contype / conaffinity off at run time removes its pairs from the next
  launch on (the device batch is rebuilt; state, time and run-time model edits carry over), and the continuation equals
  that of a Physics compiled with those bits from the start and put in the same state; switching them back restores the
  pairs."""
  full = (1 << 13) - 1      # mjSTATE_INTEGRATION: time, qpos, qvel, act, warm start and every user input
  p = make(_FILTER_XML)
  p.model.dof_damping[:] = 0.05      # a run-time edit the rebuilt batch must keep
  p.step(40)
  assert int(np.asarray(p.data.ncon).ravel()[0]) >= 3      # a on floor, b on floor, c on b
  t0, state, warm = p.data.time, p.get_state(full).copy(), np.array(p.data.qacc_warmstart)
  ia = p.model.name2id('ga', 'geom')
  p.model.geom_contype[ia] = 0
  p.named.model.geom_conaffinity['ga'] = 0
  p.step(5)
  assert abs(p.data.time - (t0 + 5 * 0.002)) < 1e-12
  q = make(_FILTER_XML.replace("name='ga'", "name='ga' contype='0' conaffinity='0'"))
  q.model.dof_damping[:] = 0.05
  q.set_state(state, full)
  q.forward()
  q.data.qacc_warmstart = warm
  q.step(5)
  np.testing.assert_allclose(np.asarray(p.data.qpos), np.asarray(q.data.qpos), rtol=0, atol=atol)
  np.testing.assert_allclose(np.asarray(p.data.qvel), np.asarray(q.data.qvel), rtol=0, atol=atol)
  z = float(np.asarray(p.named.data.xpos['a'])[2])
  p.step(200)
  assert float(np.asarray(p.named.data.xpos['a'])[2]) < z - 0.05      # a falls through the floor; b and c stay
  assert float(np.asarray(p.named.data.xpos['b'])[2]) > 0.09 and float(np.asarray(p.named.data.xpos['c'])[2]) > 0.29
  p.model.geom_contype[ia] = 1
  p.model.geom_conaffinity[ia] = 1
  with p.reset_context():
    p.data.qpos[:3] = [0, 0, 0.1]
    p.data.qvel[:6] = 0
  p.step(50)
  assert float(np.asarray(p.named.data.xpos['a'])[2]) > 0.09      # rests on the floor again
  assert int(np.asarray(p.data.ncon).ravel()[0]) >= 3


def test_collision_filter_bits_are_writable_and_rebuild_the_batch(oracle_backend):
  check_collision_filter_edits(physics_lib.Physics.from_xml_string)


# ---- data.ten_length / data.ten_velocity (locomotion/walkers/rodent.py:279-285 observes them) -----------------------------
_TENDON_XML = """<mujoco><option timestep='0.002'/><worldbody>
<body name='a' pos='0 0 1'><joint name='j1' type='hinge' axis='0 1 0'/><geom type='capsule' fromto='0 0 0 .3 0 0' size='.02'/>
 <site name='s1' pos='.3 0 0'/>
 <body name='b' pos='.3 0 0'><joint name='j2' type='hinge' axis='0 1 0'/><geom type='capsule' fromto='0 0 0 .3 0 0' size='.02'/>
  <site name='s2' pos='.3 0 .05'/></body></body>
<body name='c' pos='0 .4 1'><joint name='j3' type='slide' axis='0 0 1'/><geom type='sphere' size='.05'/><site name='s3'/></body>
<site name='s0' pos='0 0 1.5'/></worldbody>
<tendon><fixed name='f'><joint joint='j1' coef='0.6'/><joint joint='j2' coef='-0.4'/></fixed>
<spatial name='sp'><site site='s0'/><site site='s1'/><site site='s2'/></spatial>
<spatial name='sq'><site site='s2'/><site site='s3'/></spatial></tendon></mujoco>"""


def check_tendon_length_and_velocity(make, atol):
  """The facade's ten_length equals the oracle's mj_tendon, ten_velocity equals ten_J qvel, along a swinging rollout
  (fixed tendon, a two-segment spatial tendon anchored in the world, a spatial tendon between two moving bodies)."""
  from oracle.oracle import OracleModel, OraclePhysics
  p = make(_TENDON_XML)
  with p.reset_context():
    p.data.qpos[:] = [0.3, -0.5, 0.1]
    p.data.qvel[:] = [1.0, -2.0, 0.5]
  o = OraclePhysics(OracleModel(p.model))
  for k in range(4):
    o.qpos[:] = np.asarray(p.data.qpos)
    o.qvel[:] = np.asarray(p.data.qvel)
    o.forward()
    np.testing.assert_allclose(p.data.ten_length, np.asarray(o.ten_length), rtol=0, atol=atol)
    J = np.asarray(o.ten_J).reshape(p.model.ntendon, p.model.nv)
    np.testing.assert_allclose(p.data.ten_velocity, J @ np.asarray(o.qvel), rtol=0, atol=atol)
    np.testing.assert_allclose(p.named.data.ten_length['sp'], np.asarray(o.ten_length)[1], rtol=0, atol=atol)
    p.step(25)
  assert abs(p.data.ten_velocity).max() > 0.1


def test_tendon_length_and_velocity_are_served(oracle_backend):
  check_tendon_length_and_velocity(physics_lib.Physics.from_xml_string, 1e-12)
