"""CPU tier of the composer-side layer: the environment loop and hook order of
composer/environment.py:74-162,412-465, bind() semantics of mjcf/physics.py:516-652, and the task layers of BASELINE
configs 4 / 5 (tasks/go_to_target.py, soccer/task.py) on an oracle-backed stand-in of the device physics."""
import os

import numpy as np
import pytest
import torch

from dm_control_amd import composer, mjcf_compiler as mc
from dm_control_amd.composer import environment
from dm_control_amd.composer.tasks import go_to_target, soccer
from dm_control_amd.suite import common

from composer_fake import OracleDevicePhysics


def _cheetah():
  return mc.compile_xml(common.read_model('cheetah.xml'))


class _RecordingTask(environment.Task):
  physics_timestep, control_timestep = 0.01, 0.03

  def __init__(self, substep_hooks):
    self.log = []
    if substep_hooks:
      self.before_substep = lambda physics, action, rs: self.log.append('before_substep')
      self.after_substep = lambda physics, rs: self.log.append('after_substep')

  def initialize_episode(self, physics, random_state, mask):
    self.log.append('initialize_episode')

  def before_step(self, physics, action, random_state):
    self.log.append('before_step')
    physics.field('ctrl').copy_(action.T)

  def after_step(self, physics, random_state):
    self.log.append('after_step')

  def get_reward(self, physics):
    return physics.field('qvel')[0].clone()

  def get_observation(self, physics):
    return {'position': physics.field('qpos').T.clone()}


def test_trivial_hooks_are_not_dispatched_and_substeps_fuse():
  """environment.py:44-58,86-96: no-op hooks are found once and skipped; without substep hooks the n_sub_steps
  physics steps of a control step are ONE launch, with them they are n launches in the reference's order."""
  m = _cheetah()
  for hooks in (False, True):
    task = _RecordingTask(hooks)
    phys = OracleDevicePhysics(m, 2)
    env = environment.Environment(task, phys, time_limit=0.09)
    assert env.n_sub_steps == 3 and env.fused == (not hooks)
    ts = env.reset()
    assert ts.step_type.tolist() == [environment.FIRST] * 2 and task.log == ['initialize_episode']
    task.log.clear(); phys.launches.clear()
    ts = env.step(torch.zeros((2, m.nu), dtype=torch.float64))
    if hooks:
      assert task.log == ['initialize_episode', 'before_step'] + ['before_substep', 'after_substep'] * 3 + ['after_step']
      assert phys.launches == ['step1'] * 3
    else:
      assert task.log == ['initialize_episode', 'before_step', 'after_step']
      assert phys.launches == ['step3']
    assert ts.step_type.tolist() == [environment.MID] * 2
  assert environment._callable_is_trivial(environment.Task.after_step)
  assert not environment._callable_is_trivial(_RecordingTask.after_step)
  with pytest.raises(ValueError):
    env.add_extra_hook('no_such_hook', lambda *a: None)


def test_time_limit_auto_reset_per_environment():
  """environment.py:414-416,440-462: LAST at the time limit, then the next step() call returns FIRST for that
  environment only (env_mode: it is re-initialised and gets mj_forward while the others step)."""
  m = _cheetah()
  task = _RecordingTask(False)
  phys = OracleDevicePhysics(m, 3)
  env = environment.Environment(task, phys, time_limit=0.06)
  env.reset()
  phys.field('time')[0, 1] = 0.03            # environment 1 is one control step ahead
  a = torch.full((3, m.nu), 0.3, dtype=torch.float64)
  ts = env.step(a)
  assert ts.step_type.tolist() == [environment.MID, environment.LAST, environment.MID]
  q_before = phys.field('qpos').clone()
  ts = env.step(a)
  assert ts.step_type.tolist() == [environment.LAST, environment.FIRST, environment.LAST]
  assert float(ts.reward[1]) == 0.0 and float(ts.discount[1]) == 1.0
  np.testing.assert_array_equal(phys.field('qpos')[:, 1].numpy(), m.qpos0)          # re-initialised, not stepped
  assert float(phys.field('time')[0, 1]) == 0.0
  assert not np.array_equal(phys.field('qpos')[:, 0].numpy(), q_before[:, 0].numpy())
  # the reference returns reset() for it without the step hooks (environment.py:412-420): the discarded action
  # does not reach its ctrl, and the task is told which environments are restarting
  assert phys.field('ctrl')[:, 1].abs().max() == 0 and float(phys.field('ctrl')[0, 0]) == 0.3
  assert task.restarting.tolist() == [False, True, False]
  ts = env.step(a)
  assert ts.step_type.tolist() == [environment.FIRST, environment.MID, environment.FIRST]


def test_bind_resolves_elements_like_mjcf_physics():
  """mjcf/physics.py:516-652: attribute access by element kind, reads are gathers, writes are scatters."""
  m = _cheetah()
  phys = OracleDevicePhysics(m, 2, outputs=('sensordata', 'xpos', 'xmat', 'geom_xpos'))
  phys.field('qpos')[:, 1] += 0.1
  phys.forward()
  torso = phys.bind('body', 'torso')
  assert torso.element_id == m.name2id('torso', 'body')
  np.testing.assert_array_equal(torso.xpos.numpy(), phys.field('xpos')[3:6].numpy())
  feet = phys.bind('body', ['bfoot', 'ffoot'])
  assert tuple(feet.xpos.shape) == (2, 3, 2) and tuple(feet.xmat.shape) == (2, 9, 2)
  jb = phys.bind('joint', ['bthigh', 'fthigh'])
  np.testing.assert_array_equal(jb.qpos.numpy(), phys.field('qpos')[[3, 6]].numpy())
  jb.qpos = torch.tensor([0.25, -0.25], dtype=torch.float64)
  assert phys.field('qpos')[3].tolist() == [0.25, 0.25] and phys.field('qpos')[6].tolist() == [-0.25, -0.25]
  act = phys.bind('actuator', list(m.names['actuator']))
  act.ctrl = torch.ones((m.nu, 2), dtype=torch.float64) * 0.5
  assert float(phys.field('ctrl').min()) == 0.5
  np.testing.assert_allclose(phys.bind('geom', 'torso').size, m.geom_size[m.name2id('torso', 'geom')])
  with pytest.raises(AttributeError):
    torso.xpos = 0.0
  with pytest.raises(AttributeError):
    _ = torso.qpos
  with pytest.raises(ValueError):
    phys.bind('camera', 'x')


# ---- BASELINE config 4: CMU humanoid go-to-target -----------------------------------------------------
def _go_env(B, **kw):
  task = go_to_target.GoToTarget(**kw)
  phys = OracleDevicePhysics(task.model, B, outputs=('sensordata', 'xpos', 'xmat', 'contact_geom1'))
  return environment.Environment(task, phys, time_limit=30.0, random_state=3), task, phys


def _reference_observation(m, task, phys, e):
  """The enabled observables of go_to_target.py:108-118 evaluated the reference's way (per env, numpy)."""
  xpos = phys.field('xpos')[:, e].numpy().reshape(-1, 3)
  xmat = phys.field('xmat')[:, e].numpy().reshape(-1, 3, 3)
  sd = phys.field('sensordata')[:, e].numpy()
  root = m.name2id('root', 'body')
  obs = {}
  qadr = [int(m.jnt_qposadr[m.name2id(j, 'joint')]) for j in task.walker.observable_joints]
  vadr = [int(m.jnt_dofadr[m.name2id(j, 'joint')]) for j in task.walker.observable_joints]
  obs['joints_pos'] = phys.field('qpos')[:, e].numpy()[qadr]
  obs['joints_vel'] = phys.field('qvel')[:, e].numpy()[vadr]
  obs['body_height'] = xpos[root, 2:3]
  obs['world_zaxis'] = xmat[root].ravel()[6:]
  eff = np.stack([xpos[m.name2id(b, 'body')] for b in ('rradius', 'lradius', 'rfoot', 'lfoot')])
  obs['end_effectors_pos'] = np.reshape(np.dot(eff - xpos[root], xmat[root]), -1)       # cmu_humanoid.py:479-486
  obs['appendages_pos'] = np.reshape(np.dot(np.vstack([eff, xpos[m.name2id('head', 'body')]]) - xpos[root], xmat[root]), -1)
  sens = lambda n: sd[int(m.sensor_adr[m.name2id(n, 'sensor')]):][:int(m.sensor_dim[m.name2id(n, 'sensor')])]
  obs['sensors_gyro'], obs['sensors_velocimeter'] = sens('sensor_root_gyro'), sens('sensor_root_veloc')
  obs['sensors_accelerometer'] = sens('sensor_root_accel')
  obs['sensors_torque'] = np.tanh(2 * np.concatenate([sens(n) for n in task.walker.torque_sensors]) / 60)
  obs['sensors_touch'] = (np.concatenate([sens(n) for n in task.walker.touch_sensors]) > 1e-3).astype(float)
  tgt = np.append(task.target_position(phys)[:, e].numpy(), 0.0)
  obs['target'] = np.dot(tgt - xpos[root], xmat[root])
  return obs


def test_go_to_target_observations_reward_and_contact_termination():
  env, task, phys = _go_env(3)
  m = task.model
  assert (m.nq, m.nv, m.nu) == (63, 62, 56) and env.n_sub_steps == 6 and env.fused
  ts = env.reset()
  spawn = phys.field('qpos')[0:2].numpy()
  assert np.abs(spawn).max() <= 4.0 and len(set(np.round(spawn[0], 6))) == 3          # uniform over the 8 x 8 arena
  np.testing.assert_allclose(phys.field('qpos')[2:].numpy(), np.tile(m.qpos0[2:, None], (1, 3)))      # upright pose
  rs = np.random.RandomState(0)
  for t in range(4):
    ts = env.step(torch.from_numpy(rs.uniform(-1, 1, (3, m.nu))))
    for e in range(3):
      want = _reference_observation(m, task, phys, e)
      assert set(want) == set(ts.observation)
      for k, v in want.items():
        np.testing.assert_allclose(ts.observation[k][e].numpy(), v, rtol=0, atol=1e-12, err_msg=k)
    assert ts.step_type.tolist() == [environment.MID] * 3 and ts.discount.tolist() == [1.0] * 3
  # reward: 1 inside the distance tolerance of the target (go_to_target.py:176-185)
  root = phys.field('xpos')[3*m.name2id('root', 'body'):][:2]
  task._target[:, 0] = root[:, 0] + torch.tensor([0.3, -0.2], dtype=torch.float64)
  task._target[:, 1] = root[:, 1] + 5.0
  assert task.get_reward(phys).tolist()[:2] == [1.0, 0.0]
  # failure: a non-foot geom on the ground (the contact scan of :189-193) -- lay environment 2 down
  phys.field('qpos')[2, 2] = 0.12
  phys.field('qpos')[3:7, 2] = torch.tensor([1.0, 0, 0, 0], dtype=torch.float64)
  ts = env.step(torch.zeros((3, m.nu), dtype=torch.float64))
  g1 = phys.field('contact_geom1')[:, 2].numpy(); g2 = phys.field('contact_geom2')[:, 2].numpy()
  nonfoot = {m.name2id(n, 'geom') for n in task.walker.nonfoot_geoms}
  scan = any((a in nonfoot and b == 0) or (b in nonfoot and a == 0) for a, b in zip(g1, g2) if a >= 0)
  assert scan and ts.step_type.tolist() == [environment.MID, environment.MID, environment.LAST]
  assert ts.discount.tolist() == [1.0, 1.0, 0.0]
  ts = env.step(torch.zeros((3, m.nu), dtype=torch.float64))
  assert ts.step_type.tolist() == [environment.MID, environment.MID, environment.FIRST]
  assert abs(float(phys.field('qpos')[2, 2]) - m.qpos0[2]) < 1e-12          # upright again


def test_go_to_target_moving_target_counter():
  env, task, phys = _go_env(2, moving_target=True, steps_before_moving_target=2)
  m = task.model
  env.reset()
  root = lambda: phys.field('xpos')[3*m.name2id('root', 'body'):][:2]
  zero = torch.zeros((2, m.nu), dtype=torch.float64)
  task._target[:, 0] = root()[:, 0]
  task._target[:, 1] = root()[:, 1] + 6.0
  t0 = task._target.clone()
  env.step(zero)
  assert task._reward_steps.tolist() == [1, 0]
  np.testing.assert_array_equal(task._target.numpy(), t0.numpy())
  env.step(zero)                       # second rewarded step: counter reaches 2 -> after_step of the NEXT step moves it
  env.step(zero)
  assert not np.array_equal(task._target[:, 0].numpy(), t0[:, 0].numpy())
  np.testing.assert_array_equal(task._target[:, 1].numpy(), t0[:, 1].numpy())


# ---- BASELINE config 5: soccer 2v2 --------------------------------------------------------------------------
def _soccer_env(B, **kw):
  task = soccer.Soccer2v2()
  phys = OracleDevicePhysics(task.model, B, outputs=('sensordata', 'xpos', 'xmat', 'geom_xpos', 'cvel'), nconmax=24)
  return environment.Environment(task, phys, time_limit=45.0, random_state=1, **kw), task, phys


def test_soccer_substep_detectors_rewards_and_throw_in():
  env, task, phys = _soccer_env(3)
  m = task.model
  assert (m.nq, m.nv, m.nu) == (31, 30, 12) and env.n_sub_steps == 5
  # the goal detectors watch every substep (pitch.py:262) -- through the step kernel's substep probe: one launch
  assert env.fused and env.probed
  ts = env.reset()
  assert ts.observation['ball_ego_position'].shape == (3, 4, 3)
  assert ts.observation['teammate_0_ego_position'].shape == (3, 4, 3) and ts.observation['opponent_1_ego_orientation'].shape == (3, 4, 9)
  ball = task.ball_xpos(phys).numpy()
  assert np.abs(ball[0]).max() <= 24.0 and np.abs(ball[1]).max() <= 18.0 and np.allclose(ball[2], 0.5)
  a = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (3, 4, 3)))
  phys.launches.clear()
  ts = env.step(a)
  assert phys.launches == ['step5']
  np.testing.assert_array_equal(phys.field('ctrl')[:, 0].numpy(), a[0].reshape(-1).numpy())
  assert ts.reward.shape == (4, 3) and float(ts.reward.abs().max()) == 0.0
  # egocentric ball position of home0: the framepos sensor objtype = reftype = "body" of observables.py:182-186, which
  # MuJoCo evaluates between the bodies' INERTIAL frames: (xipos_ball - xipos_head) . ximat_head
  def inertial(name, e):
    b = m.name2id(name, 'body')
    pos = phys.field('xpos')[3*b:3*b + 3, e].numpy(); R = phys.field('xmat')[9*b:9*b + 9, e].numpy().reshape(3, 3)
    return pos + R @ m.body_ipos[b], R @ mc.quat_to_mat(m.body_iquat[b])
  (hp, hR), (bp, _) = inertial('home0/head_body', 1), inertial('soccer_ball/', 1)
  np.testing.assert_allclose(ts.observation['ball_ego_position'][1, 0].numpy(), (bp - hp) @ hR, atol=1e-12)
  sid = m.name2id('home0/ball_ego_pos', 'sensor')
  np.testing.assert_array_equal(ts.observation['ball_ego_position'][1, 0].numpy(),
                                phys.field('sensordata')[m.sensor_adr[sid]:m.sensor_adr[sid] + 3, 1].numpy())
  # put the ball inside the away goal in env 0 (HOME scores) and off the court in env 2
  bq = task._ball_q
  phys.field('qpos')[bq:bq + 3, 0] = torch.tensor([37.0, 0.0, 1.0], dtype=torch.float64)
  phys.field('qpos')[bq:bq + 3, 2] = torch.tensor([0.0, 27.0, 0.2], dtype=torch.float64)
  phys.field('qvel')[task._ball_v:task._ball_v + 6] = 0
  ts = env.step(torch.zeros((3, 4, 3), dtype=torch.float64))
  assert ts.reward[:, 0].tolist() == [1.0, 1.0, -1.0, -1.0] and ts.reward[:, 1].tolist() == [0.0] * 4
  assert ts.step_type.tolist() == [environment.LAST, environment.MID, environment.MID] and ts.discount.tolist() == [0.0, 1.0, 1.0]
  assert bool(task.field.detected[2]) and not bool(task.field.detected[1])
  ts = env.step(torch.zeros((3, 4, 3), dtype=torch.float64))      # env 0 restarts; env 2 gets the throw-in
  assert ts.step_type.tolist() == [environment.FIRST, environment.MID, environment.MID]
  b2 = task.ball_xpos(phys)[:, 2].numpy()
  assert abs(b2[1]) < 27.0 * 0.9 + 0.3 and not bool(task.field.detected[2])


def test_soccer_probed_launch_equals_the_per_substep_hooks():
  """A ball that crosses the goal volume INSIDE a control step (in at substep 2, out again by the last one) is a goal
  for the reference (retain_substep_detections).  The default environment sees it in the probe trace of its single
  launch exactly as `fuse_substeps=False` does with five launches and the hooks in between; `fuse_substeps=True`, which
  looks at the end of the control step only, misses it."""
  outs = {}
  xml = common.read_model('soccer_2v2_boxhead.xml').replace('<option ', '<option gravity="0 0 0" ', 1)
  assert 'gravity="0 0 0"' in xml
  model = mc.compile_xml(xml)
  model.opt.disableflags = int(model.opt.disableflags) | mc.C['DMC_DSBL_CONTACT']      # ballistic: nothing deflects the ball
  for mode in (None, False, True):
    task = soccer.Soccer2v2(model=model)
    phys = OracleDevicePhysics(task.model, 2, outputs=('sensordata', 'xpos', 'xmat', 'geom_xpos', 'cvel'), nconmax=24)
    env = environment.Environment(task, phys, time_limit=45.0, random_state=1, fuse_substeps=mode)
    env.reset()
    bq, bv = task._ball_q, task._ball_v
    # env 0: clips the upper front edge of the away goal volume (x > 34.67, z < 5.33) at (40, 0, 60) m/s: inside after
    # the first substep only; env 1: at rest mid-field
    phys.field('qpos')[bq:bq + 7, 0] = torch.tensor([34.6, 0.0, 5.0 - 0.35, 1, 0, 0, 0], dtype=torch.float64)
    phys.field('qvel')[bv:bv + 6, 0] = torch.tensor([40.0, 0.0, 60.0, 0.0, 0.0, 0.0], dtype=torch.float64)
    phys.mark_as_dirty()
    phys.launches.clear()
    ts = env.step(torch.zeros((2, 4, 3), dtype=torch.float64))
    outs[mode] = (ts.reward.clone(), ts.step_type.clone(), task.away_goal.detected.clone(), task.field.detected.clone(),
                  {k: v.clone() for k, v in ts.observation.items()}, list(phys.launches))
  assert outs[None][5] == ['step5'] and outs[False][5] == ['step1'] * 5 and outs[True][5] == ['step5']
  for a, b in zip(outs[None][:4], outs[False][:4]):
    assert torch.equal(a, b)
  for k in outs[None][4]:
    assert torch.equal(outs[None][4][k], outs[False][4][k]), k
  assert outs[None][2].tolist() == [True, False] and outs[None][1].tolist() == [environment.LAST, environment.MID]
  assert outs[None][0][:, 0].tolist() == [1.0, 1.0, -1.0, -1.0]
  assert outs[True][2].tolist() == [False, False]      # the ball has left the volume again by the end of the control step


def test_soccer_fused_checks_detectors_once_per_control_step():
  env, task, phys = _soccer_env(2, fuse_substeps=True)
  env.reset()
  phys.launches.clear()
  env.step(torch.zeros((2, 4, 3), dtype=torch.float64))
  assert phys.launches == ['step5'] and env.fused


def test_make_names():
  with pytest.raises(ValueError):
    composer.make('nope', 1)


def test_soccer_pitch_resize_rows_reproduce_the_compiled_pitch():
  """RandomizedPitch (pitch.py:645-669) on per-environment geom rows: resizing to the asset's own 40 x 30 must
  reproduce, for every wall and goal post, exactly the pose / size / bounding radius the MJCF compiler derived
  from the reference's goal-post geometry (scripts/make_soccer_model.py); another size moves walls and goals."""
  task = soccer.Soccer2v2(randomize_pitch=((32, 24), (48, 36)))
  m = task.model
  names = task.pitch_geoms()
  B = 3

  def quat2mat(q):
    w, x, y, z = q
    return np.array([[w*w+x*x-y*y-z*z, 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), w*w-x*x+y*y-z*z, 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), w*w-x*x-y*y+z*z]])
  rows = []
  for n in names:
    g = m.name2id(n, 'geom')
    rows.append(np.r_[m.geom_pos[g], quat2mat(m.geom_quat[g]).ravel(), m.geom_size[g], m.geom_rbound[g]])
  init = np.concatenate(rows)

  class _P:
    pass
  p = _P()
  p.torch, p.B, p.dtype, p.device = torch, B, torch.float64, torch.device('cpu')
  eg = torch.from_numpy(np.tile(init[:, None], (1, B))).clone()
  p.field = lambda name: eg
  p.const = lambda v: torch.from_numpy(np.asarray(v, dtype=np.float64))
  mask = torch.tensor([True, True, False])
  size = torch.tensor([[40.0, 32.0, 48.0], [30.0, 24.0, 36.0]], dtype=torch.float64)
  task._resize_pitch(p, size, mask)
  np.testing.assert_allclose(eg[:, 0].numpy(), init, rtol=0, atol=1e-12)            # same size: the compiled pitch
  np.testing.assert_array_equal(eg[:, 2].numpy(), init)                               # not in the mask: untouched
  e1 = eg[:, 1].numpy().reshape(len(names), 16)
  assert e1[0, 1] == -24.0 and e1[1, 1] == 24.0 and e1[2, 0] == -32.0 and e1[3, 0] == 32.0        # walls at the new size
  k = names.index('home_goal/top_post')
  assert abs(e1[k, 13] - 24 * 0.33) < 1e-12 and abs(e1[k, 0] - (-32.0 + 32. / 6.)) < 1e-12       # half-length, goal line
  lo, hi = task.home_goal.bounds(p)
  assert abs(float(lo[0, 1]) + 32.0) < 1e-12 and abs(float(hi[1, 1]) - 24 * 0.33) < 1e-12
  flo, fhi = task.field.bounds(p)
  assert abs(float(fhi[0, 1]) - (32.0 - 32. / 6.)) < 1e-12 and float(fhi[0, 2]) == 40.0 - 32. / 6.      # masked-out env keeps 40 x 30


def test_soccer_task_kernels_build_and_agree_on_their_argument_struct():
  """tasks/soccer_task.hip cross-compiles for gfx950 (no GPU needed) and its SoccerArgs has the size of the ctypes
  mirror in tasks/soccer.py, for both precisions (task_kernels() raises on a mismatch); the entry points exist."""
  import ctypes
  for precision in (32, 64):
    lib, S = soccer.task_kernels(precision)
    assert lib.soccer_args_size() == ctypes.sizeof(S)
    assert hasattr(lib, 'soccer_pre') and hasattr(lib, 'soccer_post')
  env, _, _ = _soccer_env(2)
  # the CPU stand-in has no device tensors: the environment runs the hooks
  assert env.task.device_step(env, None) is None
