"""GPU tier of the composer-side layer: the device observation gather, per-env launch modes, and the BASELINE
config 4 / 5 environments against a host evaluation of the same state (the reference's per-env numpy formulas)."""
import numpy as np
import pytest

from dm_control_amd import mjcf_compiler as mc, observation
from dm_control_amd.suite import common

pytestmark = pytest.mark.gpu


def _model(name):
  return mc.compile_xml(common.read_model(name + '.xml'))


@pytest.mark.parametrize('precision', [32, 64])
def test_device_gather_matches_host_gather(precision):
  """dmc_gather_* (composer/observation/updater.py:285-295 for a batch): one launch, (B, nobs) env-major, with the
  corruptors of the walkers' touch / torque observables, against the host evaluation of the same table."""
  import torch
  from dm_control_amd.batch import BatchedPhysics
  from dm_control_amd.physics import Physics
  m = _model('humanoid')
  B = 77                                   # not a multiple of the 64-env tile
  b = BatchedPhysics(m, B, precision=precision, nconmax=24)
  rs = np.random.RandomState(0)
  q = np.tile(m.qpos0, (B, 1)); q[:, 7:] += rs.uniform(-.3, .3, (B, m.nq - 7)); q[:, 2] = rs.uniform(0.2, 1.5, B)
  b.set('qpos', q); b.set('qvel', rs.uniform(-1, 1, (B, m.nv)))
  b.forward()
  touch = [n for i, n in enumerate(m.names['sensor']) if m.sensor_type[i] == 0]
  table = observation.GatherTable(m, [
      ('qpos', ['right_knee', 'left_knee', 'root']), ('qvel', None), ('xpos', ['head', 'torso'], 'z'),
      ('xmat', ['torso'], ['zx', 'zy', 'zz']), ('sensordata', touch, None, ('greater', 1e-3)),
      ('sensordata', ['torso_accel'], None, ('tanh2', 60.0)), ('sensordata', ['torso_gyro'], None, ('asinh', 0.0)),
      ('subtree_com', ['torso']), ('sensordata', touch, None, ('log1p', 0.0))] * 3)       # > 64 rows: several tiles
  assert table.size > 128
  dg = table.on_device(b)
  out = dg()
  torch.cuda.synchronize()
  assert tuple(out.shape) == (B, table.size)

  class _Host:      # the host mirror's surface GatherTable.gather needs
    batch_size = B
    batch = b
  want = table.gather(_Host)
  tol = 1e-12 if precision == 64 else 2e-6
  np.testing.assert_allclose(out.cpu().numpy(), want, rtol=tol, atol=tol)
  # a rebound field is followed
  qdev = torch.from_numpy(np.ascontiguousarray((q + 1.0).T)).to('cuda').to(out.dtype)
  b.sync(); b.bind('qpos', qdev.data_ptr())
  out2 = dg()
  np.testing.assert_allclose(out2[:, 0].cpu().numpy(), q[:, m.jnt_qposadr[m.name2id('right_knee', 'joint')]] + 1.0, rtol=tol, atol=tol)
  with pytest.raises(Exception):
    observation.GatherTable(m, [('qpos', ['no_such_joint'])])
  dg.close(); b.close()


def test_env_mode_per_environment_launch_override():
  """"env_mode": 0 = step, 1 = mj_forward with actuation disabled instead (reset_context while the batch steps),
  2 = untouched."""
  from dm_control_amd.batch import BatchedPhysics, OUT
  m = _model('hopper')
  B = 6
  rs = np.random.RandomState(1)
  q = np.tile(m.qpos0, (B, 1)); q[:, 1] += 0.3; q[:, 3:] += rs.uniform(-.2, .2, (B, m.nq - 3))
  c = rs.uniform(-1, 1, (B, m.nu))
  a, b = BatchedPhysics(m, B, precision=64), BatchedPhysics(m, B, precision=64)
  for e in (a, b):
    e.set('qpos', q); e.set_control(c); e.set_output_mask(OUT['sensor'] | OUT['xpos'] | OUT['actuator'])
    e.step(3)
  mode = np.array([0, 1, 2, 0, 2, 1], dtype=np.int32)[:, None]
  before = {f: b.get(f) for f in ('qpos', 'qvel', 'xpos', 'sensordata', 'time', 'actuator_force')}
  b.set('env_mode', mode)
  a.step(2); b.step(2)
  fwd = BatchedPhysics(m, B, precision=64)
  fwd.set('qpos', before['qpos']); fwd.set('qvel', before['qvel']); fwd.set_control(c)
  fwd.set('qacc_warmstart', a.get('qacc_warmstart') * 0 + b.get('qacc_warmstart'))
  fwd.forward(disable_actuation=True)
  for e in range(B):
    if mode[e, 0] == 0:
      for f in before:
        np.testing.assert_array_equal(b.get(f)[e], a.get(f)[e], err_msg='%s env %d' % (f, e))
    else:
      for f in ('qpos', 'qvel', 'time'):
        np.testing.assert_array_equal(b.get(f)[e], before[f][e], err_msg='%s env %d' % (f, e))
      if mode[e, 0] == 2:
        for f in ('xpos', 'sensordata', 'actuator_force'):
          np.testing.assert_array_equal(b.get(f)[e], before[f][e])
      else:
        np.testing.assert_allclose(b.get('xpos')[e], fwd.get('xpos')[e], atol=1e-13)
        assert np.abs(b.get('actuator_force')[e]).max() == 0.0 and np.abs(before['actuator_force'][e]).max() > 0
  a.close(); b.close(); fwd.close()


def test_go_to_target_environment_on_gpu():
  """BASELINE config 4 as an environment (tasks/go_to_target.py): observations / reward / contact-scan termination
  of the device task against the per-env host evaluation of the same device state; auto-reset without host sync."""
  import torch
  from dm_control_amd import composer
  from dm_control_amd.composer import environment
  from test_composer_cpu import _reference_observation
  B = 48
  env = composer.make('cmu_go_to_target', B, random_state=5)
  task, phys, m = env.task, env.physics, env.task.model
  assert env.fused and env.n_sub_steps == 6
  ts = env.reset()
  gen = torch.Generator(device='cuda').manual_seed(0)
  nonfoot = {m.name2id(n, 'geom') for n in task.walker.nonfoot_geoms}
  seen_last = seen_first = 0
  launches0 = env.launches
  for t in range(60):
    a = torch.rand((B, m.nu), device='cuda', generator=gen) * 2 - 1
    prev_reset = env._reset_next.clone()
    ts = env.step(a)
    torch.cuda.synchronize()

    class _Host:      # CPU copies of the device fields for the per-env reference formulas
      B = 0
      def __init__(s): s.f = {n: phys.field(n).cpu().double() for n in ('qpos', 'qvel', 'xpos', 'xmat', 'sensordata')}
      def field(s, n): return s.f[n]
    host = _Host()
    tgt = task.target_position(phys).cpu().double()
    task_view = type('T', (), {'walker': task.walker, 'target_position': staticmethod(lambda p: tgt)})
    g1 = phys.field('contact_geom1').cpu().numpy(); g2 = phys.field('contact_geom2').cpu().numpy()
    for e in (0, 7, B - 1):
      want = _reference_observation(m, task_view, host, e)
      for k, v in want.items():
        np.testing.assert_allclose(ts.observation[k][e].cpu().numpy(), v, rtol=0, atol=2e-5, err_msg='%s step %d' % (k, t))
    # termination = the reference's contact scan (go_to_target.py:189-193), per env in Python
    scan = np.array([any((x in nonfoot and y == 0) or (y in nonfoot and x == 0) for x, y in zip(g1[:, e], g2[:, e]) if x >= 0)
                     for e in range(B)])
    first = prev_reset.cpu().numpy()
    st = ts.step_type.cpu().numpy()
    assert (st[first] == environment.FIRST).all()
    assert ((st == environment.LAST) == (scan & ~first)).all(), t
    assert (ts.discount.cpu().numpy()[scan & ~first] == 0).all()
    root = phys.field('xpos')[3*m.name2id('root', 'body'):][:2].cpu().numpy()
    want_r = (np.linalg.norm(tgt.numpy() - root, axis=0) < 1.0) & ~first
    np.testing.assert_array_equal(ts.reward.cpu().numpy() > 0, want_r)
    seen_last += int((st == environment.LAST).sum()); seen_first += int(first.sum())
  assert env.launches - launches0 == 60              # one launch per control step, resets included
  assert seen_last > 0 and seen_first > 0            # random actions make the humanoid fall: episodes end and restart
  assert int(phys.field('warning').sum()) == 0
  env.close()


def test_soccer_environment_on_gpu():
  """BASELINE config 5 as an environment (tasks/soccer.py): per-substep goal detectors, 4-agent rewards, throw-in."""
  import torch
  from dm_control_amd import composer
  from dm_control_amd.composer import environment
  B = 16
  env = composer.make('soccer_2v2', B, random_state=2)
  task, phys, m = env.task, env.physics, env.task.model
  assert env.fused and env.probed and env.n_sub_steps == 5      # the goal detectors read the kernel's substep probe
  ts = env.reset()
  assert tuple(ts.observation['ball_ego_position'].shape) == (B, 4, 3)
  gen = torch.Generator(device='cuda').manual_seed(0)
  l0 = env.launches
  for t in range(10):
    ts = env.step(torch.rand((B, 4, 3), device='cuda', generator=gen) * 2 - 1)
  assert env.launches - l0 == 10
  torch.cuda.synchronize()
  # the framepos sensor objtype = reftype = "body" (observables.py:182-186): between the bodies' INERTIAL frames
  from dm_control_amd import mjcf_compiler

  def inertial(name, e):
    b = m.name2id(name, 'body')
    pos = phys.field('xpos')[3*b:3*b + 3, e].cpu().double().numpy()
    R = phys.field('xmat')[9*b:9*b + 9, e].cpu().double().numpy().reshape(3, 3)
    return pos + R @ m.body_ipos[b], R @ mjcf_compiler.quat_to_mat(m.body_iquat[b])
  (hp, hR), (bp, _) = inertial('away1/head_body', 3), inertial('soccer_ball/', 3)
  np.testing.assert_allclose(ts.observation['ball_ego_position'][3, 3].cpu().numpy(), (bp - hp) @ hR, atol=2e-5)
  assert float(ts.reward.abs().max()) == 0.0
  bq = task._ball_q
  q = phys.field('qpos')
  q[bq:bq + 3, 0] = torch.tensor([-37.0, 0.0, 1.0], device='cuda', dtype=q.dtype)       # in the home goal: AWAY scores
  q[bq:bq + 3, 5] = torch.tensor([0.0, -27.5, 0.2], device='cuda', dtype=q.dtype)       # off the court
  phys.field('qvel')[task._ball_v:task._ball_v + 6] = 0
  phys.mark_as_dirty()
  ts = env.step(torch.zeros((B, 4, 3), device='cuda'))
  assert ts.reward[:, 0].tolist() == [-1.0, -1.0, 1.0, 1.0]
  st = ts.step_type.cpu().numpy()
  assert st[0] == environment.LAST and (st[1:] == environment.MID).all() and float(ts.discount[0]) == 0.0
  assert bool(task.field.detected[5])
  ts = env.step(torch.zeros((B, 4, 3), device='cuda'))
  assert int(ts.step_type[0]) == environment.FIRST and not bool(task.field.detected[5])
  assert int(phys.field('warning').sum()) == 0
  env.close()


@pytest.mark.parametrize('precision', [64, 32])
def test_soccer_probed_launch_equals_the_per_substep_hooks_on_gpu(precision):
  """dmc_batch_set_step_probe: one fused launch whose substep probe feeds the goal / out-of-court detectors against
  n_sub_steps launches with the after_substep hooks in between (`fuse_substeps=False`): same physics (step(5) == 5 x
  step()), same detections, rewards, step types and observations over episodes with goals, throw-ins and auto-resets --
  and a ball that is inside the goal volume for one substep only is a goal for both."""
  import torch
  from dm_control_amd import composer
  from dm_control_amd.composer import environment
  B = 24
  envs = [composer.make('soccer_2v2', B, random_state=6, precision=precision, fuse_substeps=f, task_kernels=False) for f in (None, False)]      # (both as tensor operations: the task kernels have their own test below)
  assert envs[0].probed and not envs[1].fused
  gen = torch.Generator(device='cuda').manual_seed(3)
  acts = torch.rand((30, B, 4, 3), device='cuda', generator=gen) * 2 - 1
  for e in envs:
    e.reset()
  dt = envs[0].physics.dtype
  scored, worst = 0, 0.0
  for t in range(30):
    if t in (3, 11):      # kick the ball towards a goal in a few environments: it crosses the volume within the control step
      for e in envs:
        task, p = e.task, e.physics
        bq, bv = task._ball_q, task._ball_v
        p.field('qpos')[bq:bq + 7, :4] = torch.tensor([33.0, 0.0, 1.0, 1, 0, 0, 0], device='cuda', dtype=dt)[:, None]
        p.field('qvel')[bv:bv + 6, :4] = torch.tensor([60.0, 0.0, 2.0, 0, 0, 0], device='cuda', dtype=dt)[:, None]
        p.mark_as_dirty()
    out = [e.step(acts[t]) for e in envs]
    torch.cuda.synchronize()
    a, b = out
    assert torch.equal(a.step_type, b.step_type), t
    assert torch.equal(a.reward, b.reward) and torch.equal(a.discount, b.discount), t
    for k in a.observation:
      if precision == 64:
        assert torch.equal(a.observation[k], b.observation[k]), (k, t)
      else:
        # fp32: the fused launch and the single-step launches run different instantiations of the stages (full / trailing
        # partial), whose fused-multiply-add contraction the compiler is free to choose differently: last-bit
        # differences in the kinematics, not bit-equality
        err = float((a.observation[k] - b.observation[k]).abs().max())
        worst = max(worst, err)
        assert err <= 2e-3 * max(1.0, float(b.observation[k].abs().max())), (k, t, err)
    scored += int((a.step_type == environment.LAST).sum())
  print('measured: soccer probed vs hooked launches, fp%d: max observation difference %.2e over 30 control steps' % (precision, worst))
  assert scored >= 4
  assert envs[0].launches == 1 + 30 and envs[1].launches == 1 + 150      # one launch per control step against five
  for e in envs:
    assert int(e.physics.field('warning')[:8].sum()) == 0
    e.close()


@pytest.mark.parametrize('precision', [64, 32])
def test_soccer_task_kernels_equal_the_tensor_task_layer(precision):
  """tasks/soccer_task.hip (`Soccer2v2.device_step`: the control step as six launches) against the task's hooks as tensor
  operations (`task_kernels=False`, ~150 launches), from the same seeds over episodes with goals, throw-ins, time-limit
  ends and auto-resets: same placements and throw-ins (the same random stream, bit-equal states), step types, rewards,
  discounts; observations equal to rounding (the einsum / norm reductions of the tensor form may sum in another order)."""
  import torch
  from dm_control_amd import composer
  from dm_control_amd.composer import environment
  B = 40
  envs = [composer.make('soccer_2v2', B, random_state=9, precision=precision, time_limit=0.4, task_kernels=k) for k in (True, False)]
  gen = torch.Generator(device='cuda').manual_seed(5)
  T = 48
  acts = torch.rand((T, B, 4, 3), device='cuda', generator=gen) * 2 - 1
  for e in envs:
    e.reset()
  dt = envs[0].physics.dtype
  ends, throw_ins, worst = 0, 0, 0.0
  for t in range(T):
    for e in envs:
      task, p = e.task, e.physics
      bq, bv = task._ball_q, task._ball_v
      if t in (3, 21):      # a shot that crosses the goal volume within the control step, in a few environments
        p.field('qpos')[bq:bq + 7, :4] = torch.tensor([33.0, 0.0, 1.0, 1, 0, 0, 0], device='cuda', dtype=dt)[:, None]
        p.field('qvel')[bv:bv + 6, :4] = torch.tensor([60.0, 0.0, 2.0, 0, 0, 0], device='cuda', dtype=dt)[:, None]
        p.mark_as_dirty()
      if t in (5, 30):      # balls rolling off the court: thrown in at the next control step
        p.field('qpos')[bq:bq + 3, 8:14] = torch.tensor([0.0, -27.5, 0.2], device='cuda', dtype=dt)[:, None]
        p.field('qvel')[bv:bv + 6, 8:14] = 0
        p.mark_as_dirty()
    if t in (6, 31):
      throw_ins += int(envs[0].task.field.detected.sum())
      assert torch.equal(envs[0].task.field.detected, envs[1].task.field.detected)
    a, b = [e.step(acts[t]) for e in envs]
    torch.cuda.synchronize()
    assert torch.equal(a.step_type, b.step_type), t
    assert torch.equal(a.reward, b.reward) and torch.equal(a.discount, b.discount), t
    for f in ('qpos', 'qvel', 'ctrl'):
      assert torch.equal(envs[0].physics.field(f), envs[1].physics.field(f)), (f, t)
    assert torch.equal(envs[0].task.detectors.state, envs[1].task.detectors.state), t
    assert set(a.observation) == set(b.observation)
    for k in a.observation:
      assert a.observation[k].shape == b.observation[k].shape, k
      err = float((a.observation[k] - b.observation[k]).abs().max() / max(1.0, float(b.observation[k].abs().max())))
      worst = max(worst, err)
      assert err <= (1e-12 if precision == 64 else 2e-6), (k, t, err)
    ends += int((a.step_type == environment.LAST).sum())
  print('measured: soccer task kernels vs tensor task layer, fp%d: max relative observation difference %.2e over %d control steps, '
        '%d episode ends, %d throw-ins' % (precision, worst, T, ends, throw_ins))
  assert envs[0].task._dev is not None and '_dev' not in envs[1].task.__dict__
  assert ends >= B + 4 and throw_ins >= 6      # every environment ran into the time limit at least once; goals; throw-ins
  assert envs[0].launches == envs[1].launches == 1 + T
  for e in envs:
    assert int(e.physics.field('warning')[:8].sum()) == 0
    e.close()


@pytest.mark.parametrize('precision,tol', [(64, 1e-9), (32, 2e-4)])
def test_per_env_world_geoms_two_pitch_sizes_in_one_launch(precision, tol):
  """Per-environment model deltas (dmc_batch_set_env_geoms; soccer/pitch.py:612-690 RandomizedPitch): a batch whose
  environments have DIFFERENT wall positions and goal-post sizes, stepped in one launch, matches per-environment
  oracles whose models were edited accordingly."""
  from dm_control_amd.batch import BatchedPhysics
  from oracle.oracle import OraclePhysics
  m = _model('soccer_2v2_boxhead')
  from dm_control_amd.composer.tasks import soccer
  adr = soccer.addresses(m)
  bq, bv = adr['ball_q'], adr['ball_v']
  names = ['//unnamed_geom_%d' % k for k in (1, 2, 3, 4)] + ['home_goal/right_post', 'away_goal/top_post']      # the four wall planes are unnamed in pitch.py:410-420
  B = 6
  scales = np.array([0.3, 0.27, 0.33, 0.3, 0.27, 0.36])
  b = BatchedPhysics(m, B, precision=precision, nconmax=24)
  b.set_env_geoms(names)
  rs = np.random.RandomState(0)
  refs = [OraclePhysics(m) for _ in range(B)]
  for n in names:
    g = m.name2id(n, 'geom')
    wall = n.startswith('//unnamed_geom')
    pos = np.array([np.array(m.geom_pos[g]) * (scales[e] if wall else 1.0) + (0 if wall else rs.uniform(-1, 1, 3) * [2, 2, 0]) for e in range(B)])
    size = np.tile(np.array(m.geom_size[g]) * (1.0 if wall else 1.5), (B, 1))
    b.set_env_geom(n, pos=pos, size=size)
    rows = b.pack_env_geom(n, pos, m.geom_quat[g], size)
    for e, o in enumerate(refs):
      o.model.field('geom_pos')[3*g:3*g + 3] = pos[e]
      o.model.field('geom_size')[3*g:3*g + 3] = size[e]
      o.model.field('geom_rbound')[g] = rows[e, 15]
  q = np.tile(soccer.kickoff_qpos(m), (B, 1)); q[:, bq:bq + 2] = (6.0, 3.0)
  v = np.zeros((B, m.nv)); v[:, bv:bv + 3] = (40.0, 25.0, 1.0)
  b.set('qpos', q); b.set('qvel', v)
  for e, o in enumerate(refs):
    o.qpos[:] = q[e]; o.qvel[:] = v[e]
    o.forward()
  hit = np.zeros(B, bool)
  for t in range(300):
    c = rs.uniform(-1, 1, (B, m.nu))
    if precision == 32 and t:
      b.set('qpos', np.stack([o.qpos for o in refs])); b.set('qvel', np.stack([o.qvel for o in refs]))
      b.set('qacc_warmstart', np.stack([o.qacc_warmstart for o in refs]))
    b.set_control(c)
    b.step()
    for e, o in enumerate(refs):
      o.ctrl[:] = c[e]
      o.step()
      for k in range(o.ncon):
        if m.names['geom'][o.contact(k)['geom1']].startswith('//unnamed_geom'):
          hit[e] = True
    if precision == 32 and t < 3:
      continue
    qo = np.stack([o.qpos for o in refs])
    np.testing.assert_allclose(b.get('qpos'), qo, rtol=0, atol=tol * max(1.0, np.abs(qo).max()), err_msg='step %d' % t)
  assert hit.all() and not b.get('warning').any()
  ball = b.get('qpos')[:, bq:bq + 2]
  assert (np.abs(ball[:, 0]) < 40 * scales + 1).all() and len(set(np.round(ball[:, 0], 3))) > 2       # different pitches, different games
  b.close()


def test_soccer_randomized_pitch_on_gpu():
  """soccer.load's RandomizedPitch ((32, 24) .. (48, 36), soccer/__init__.py:140-148) per environment: every env
  plays on its own pitch -- walls contain its ball, goals sit on its goal lines -- and a reset draws a new one."""
  import torch
  from dm_control_amd import composer
  B = 24
  env = composer.make('soccer_2v2', B, random_state=4, randomize_pitch=((32, 24), (48, 36)), fuse_substeps=True)
  task, phys = env.task, env.physics
  env.reset()
  size = task._size_t.cpu().numpy()
  assert (size[0] >= 32).all() and (size[0] <= 48).all() and (size[1] >= 24).all() and (size[1] <= 36).all()
  assert len(set(np.round(size[0], 3))) > B // 2
  eg = phys.field('env_geom').cpu().numpy().reshape(24, 16, B)
  np.testing.assert_allclose(eg[0, 1], -size[1], rtol=1e-6); np.testing.assert_allclose(eg[3, 0], size[0], rtol=1e-6)
  # shoot every ball along +x beside the goal mouth: it leaves ITS field (the inverted field detector of its own pitch,
  # size_x - 2 goal_depth) and is thrown back in (task.py:215-217) -- it never gets further than that line plus
  # one control step of travel, and the per-env walls / posts produce no spurious contact on the way
  q, v = phys.field('qpos'), phys.field('qvel')
  bq, bv = task._ball_q, task._ball_v
  q[bq] = 0.0
  q[bq + 1] = 0.8 * (task._size_t[1] - 32. / 6.)      # inside every env's own field, beside its goal mouth
  q[bq + 2] = 0.3
  v[bv:bv + 6] = 0
  v[bv] = 60.0
  phys.mark_as_dirty()
  xmax = torch.zeros(B, device='cuda', dtype=q.dtype)
  thrown = torch.zeros(B, device='cuda', dtype=torch.bool)
  for t in range(60):
    ts = env.step(torch.zeros((B, 4, 3), device='cuda'))
    xmax = torch.maximum(xmax, task.ball_xpos(phys)[0])
    thrown |= task.field.detected
  xmax = xmax.cpu().numpy()
  line = size[0] - 2 * (32. / 6. / 2)
  assert thrown.all() and (xmax >= line - 0.05).all() and (xmax <= line + 1.6).all(), (xmax - line)
  assert int(phys.field('warning').sum()) == 0
  # a reset draws a new pitch for the environments that restart
  task.home_goal.detected[:] = False
  old = size.copy()
  env._reset_next[:5] = True
  env.step(torch.zeros((B, 4, 3), device='cuda'))
  new = task._size_t.cpu().numpy()
  assert (new[:, 5:] == old[:, 5:]).all() and (new[:, :5] != old[:, :5]).any()
  env.close()


@pytest.mark.parametrize('name,shape', [('soccer_2v2', (4, 3)), ('cmu_go_to_target', None)])
def test_environment_step_as_hip_graph(name, shape):
  """Environment.capture / step_graph: the whole control step (hooks, physics launches, reward, termination, observation
  gather) replayed as ONE HIP graph produces the same time steps as the eager loop, including auto-resets."""
  import torch
  from dm_control_amd import composer
  B = 32
  envs = [composer.make(name, B, random_state=9) for _ in range(2)]
  m = envs[0].task.model
  ashape = (B,) + (shape if shape else (m.nu,))
  gen = torch.Generator(device='cuda').manual_seed(1)
  acts = torch.rand((40,) + ashape, device='cuda', generator=gen) * 2 - 1
  eager, graph = envs
  eager.reset(); graph.reset()
  for t in range(2):                       # the warm-up steps capture() takes, mirrored on the eager environment
    eager.step(acts[0])
  graph.capture(acts[0])                   # (capturing records the step, it does not run it)
  l0 = graph.launches
  n_last = 0
  for t in range(1, 40):
    a = eager.step(acts[t])
    b = graph.step_graph(acts[t])
    torch.cuda.synchronize()
    assert torch.equal(a.step_type, b.step_type), t
    assert torch.equal(a.reward, b.reward) and torch.equal(a.discount, b.discount)
    for k in a.observation:
      assert torch.equal(a.observation[k], b.observation[k]), (k, t)
    n_last += int((a.step_type == composer.LAST).sum())
  assert graph.launches == l0              # no Python-side launch happened during the replays
  if name == 'cmu_go_to_target':
    assert n_last > 0                      # falls happened and the graph re-initialised those environments itself
  for e in envs:
    e.close()


def test_invalidation_inside_a_captured_graph_reaches_the_kernel():
  """The stash epoch lives in device memory (StepIO::epoch) and `mark_as_dirty` bumps it on the current stream
  (dmc_batch_invalidate_async): a captured sequence [edit qvel through the bound tensor, mark_as_dirty, step] replays
  with its invalidation.  With the FULL position / velocity-stage stash on (option 'stash'), a by-value epoch frozen
  at capture time made every replay continue from the stage stashed before the edit."""
  import torch
  from dm_control_amd.composer.physics import DevicePhysics
  m = _model('cheetah')
  B = 64
  rs = np.random.RandomState(1)
  q0 = np.tile(m.qpos0, (B, 1)); q0[:, 3:] += rs.uniform(-.3, .3, (B, m.nq - 3))
  acts = torch.as_tensor(rs.uniform(-1, 1, (12, m.nu, B)), device='cuda', dtype=torch.float32)
  kick = torch.as_tensor(rs.uniform(-2, 2, (m.nv, B)), device='cuda', dtype=torch.float32)
  out = []
  for mode in ('eager', 'graph'):
    phys = DevicePhysics(m, B, precision=32)
    phys.batch.set_opt('stash', 1)
    phys.field('qpos').copy_(torch.as_tensor(q0.T, device='cuda', dtype=torch.float32))
    phys.mark_as_dirty()
    phys.forward()
    ctrl = phys.field('ctrl')

    def body():
      phys.field('qvel').add_(kick)      # the throw-in of the advisor's scenario: state edited through a tensor
      phys.mark_as_dirty()
      phys.step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for t in range(2):
        ctrl.copy_(acts[t]); body()
    torch.cuda.current_stream().wait_stream(side)
    if mode == 'graph':
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
        body()
    for t in range(2, 12):
      ctrl.copy_(acts[t])
      g.replay() if mode == 'graph' else body()
    torch.cuda.synchronize()
    out.append(phys.field('qpos').clone())
    assert phys.batch.info()['stash'] == 1
    phys.close()
  assert torch.equal(out[0], out[1])


@pytest.mark.gpu
@pytest.mark.parametrize('name,shape,hist,delayed,pad', [('cmu_go_to_target', None, 'joints_pos', 'sensors_touch', 'zero'),
                                                        ('cmu_go_to_target', None, 'joints_pos', 'sensors_touch', 'initial_value'),
                                                        ('soccer_2v2', (4, 3), 'joints_vel', 'ball_ego_position', 'zero')])
def test_buffered_and_delayed_observations_on_the_device(name, shape, hist, delayed, pad):
  """composer/updater.py on the device: a history of three control steps and a one-control-step delay, per-environment
  episode clocks (go-to-target walkers fall at different steps), eager and as a replayed HIP graph.  Row i of the history
  at control step t is the plain environment's observation of step t - 2 + i (zeros before the episode began); the
  delayed observable is the previous control step's; the control step stays one physics launch."""
  import torch
  from dm_control_amd import composer
  B = 32
  kw = dict(task_kernels=False) if name == 'soccer_2v2' else {}
  plain = composer.make(name, B, random_state=9, **kw)
  n = plain.n_sub_steps
  opts = {hist: dict(update_interval=n, buffer_size=3), delayed: dict(update_interval=n, delay=n)}
  eager = composer.make(name, B, random_state=9, observation_options=opts, delayed_observation_padding=pad, **kw)
  graph = composer.make(name, B, random_state=9, observation_options=opts, delayed_observation_padding=pad, **kw)
  m = plain.task.model
  ashape = (B,) + (shape if shape else (m.nu,))
  gen = torch.Generator(device='cuda').manual_seed(1)
  acts = torch.rand((40,) + ashape, device='cuda', generator=gen) * 2 - 1
  seen = [plain.reset()]
  assert hist in seen[0].observation and delayed in seen[0].observation
  eager.reset(); graph.reset()
  for t in range(2):
    seen.append(plain.step(acts[0])); eager.step(acts[0])
  graph.capture(acts[0])
  age = torch.full((B,), 2, dtype=torch.int64, device='cuda')
  restarts = 0
  first_obs = {k: seen[0].observation[k].clone() for k in (hist, delayed)}      # the episode's first sample per environment
  for t in range(1, 40):
    p = plain.step(acts[t]); seen.append(p)
    a = eager.step(acts[t]); b = graph.step_graph(acts[t])
    torch.cuda.synchronize()
    assert eager.launches == plain.launches
    assert torch.equal(a.step_type, p.step_type) and torch.equal(a.reward, p.reward)
    started = p.step_type == composer.FIRST
    age = torch.where(started, torch.zeros_like(age), age + 1)
    restarts += int(started.sum())
    for k in first_obs:
      first_obs[k] = torch.where(started.reshape((B,) + (1,) * (first_obs[k].dim() - 1)), p.observation[k], first_obs[k])
    padding = (lambda k, like: torch.zeros_like(like)) if pad == 'zero' else (lambda k, like: first_obs[k])
    h = a.observation[hist]
    assert h.shape[1] == 3 and h.shape[0] == B
    for i in range(3):
      back = 2 - i
      old = seen[-1 - back].observation[hist]
      live = (age >= back).reshape((B,) + (1,) * (old.dim() - 1))
      assert torch.equal(h[:, i], torch.where(live, old, padding(hist, old))), (t, i)
    old = seen[-2].observation[delayed]
    live = (age >= 1).reshape((B,) + (1,) * (old.dim() - 1))
    assert torch.equal(a.observation[delayed], torch.where(live, old, padding(delayed, old))), t
    for k in p.observation:
      if k not in (hist, delayed):
        assert torch.equal(a.observation[k], p.observation[k]), (k, t)
      assert torch.equal(a.observation[k], b.observation[k]), (k, t)      # the replayed graph = the eager loop
  if name == 'cmu_go_to_target':
    assert restarts > 0
  for e in (plain, eager, graph):
    e.close()
