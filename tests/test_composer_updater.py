"""dm_control_amd.composer.updater (buffered / delayed / aggregated observations of a batch, closed form on tensors) against
the reference's own `observation.Updater` + `obs_buffer.Buffer` (composer/observation/updater.py, obs_buffer.py, executed
unmodified from the reference tree: tests/reference_pymjcf.py), one reference updater per environment, value for value:
a grid of update intervals, delays, buffer sizes, control-step lengths, paddings and aggregators, with environments that
restart at different control steps (their episode clocks then differ).  Then the same through `composer.Environment` on
the CPU stand-in of the device physics (tests/composer_fake.py)."""
import itertools
import sys

import numpy as np
import pytest

import reference_pymjcf as rp

pytestmark = pytest.mark.skipif(not rp.available(), reason='reference tree not present')
torch = pytest.importorskip('torch')

from dm_control_amd.composer import updater as bu  # noqa: E402


@pytest.fixture(scope='module')
def ref():
  rp.load()
  yield sys.modules['dm_control.composer.observation.updater'], sys.modules['dm_control.composer.observation.observable.base']
  rp.unload()


class _NoPhysics:
  """What the reference's updater needs of a physics when the observables do not look at one."""

  def suppress_physics_errors(self):
    import contextlib
    return contextlib.nullcontext()


class _World:
  """What the observables look at: the number of physics steps taken so far.  value(e, t) is unique per (env, time)."""

  def __init__(self, B):
    self.t = np.zeros(B, dtype=np.int64)

  def value(self, e):
    return 1000.0 * e + self.t[e] + np.array([0.0, 0.25, 0.5])


RESTARTS = {1: (3, 4, 9), 2: (5,), 3: (1, 2, 3, 12)}      # env -> control steps at which its episode restarts (env 0: never)


def _run_reference(ref, B, n, C, opts, pad_initial, strip):
  upd_mod, obs_mod = ref
  world = _World(B)
  out = [[] for _ in range(B)]
  ups = []
  for e in range(B):
    ob = obs_mod.Generic(lambda physics, e=e: world.value(e), update_interval=opts['update_interval'],
                         buffer_size=opts['buffer_size'], delay=opts['delay'], aggregator=opts['aggregator'])
    ob.enabled = True
    u = upd_mod.Updater({'x': ob}, physics_steps_per_control_step=n, strip_singleton_buffer_dim=strip,
                        pad_with_initial_value=pad_initial)
    u.reset(_NoPhysics(), None)
    ups.append(u)
    out[e].append(np.array(u.get_observation()['x']))
  for c in range(1, C):
    for e in range(B):
      u = ups[e]
      if c in RESTARTS.get(e, ()):
        world.t[e] += n      # (the batch's clock: a restarting environment's state is re-initialised, the others step)
        u.reset(_NoPhysics(), None)
      else:
        u.prepare_for_next_control_step()
        for _ in range(n):
          world.t[e] += 1
          u.update()
      out[e].append(np.array(u.get_observation()['x']))
  return out


def _run_batched(B, n, C, opts, pad_initial, strip):
  world = _World(B)
  values = lambda: {'x': torch.from_numpy(np.stack([world.value(e) for e in range(B)]))}
  u = bu.Updater(torch, B, n, {'x': opts}, pad=bu.PAD_INITIAL if pad_initial else bu.PAD_ZERO, strip_singleton_buffer_dim=strip)
  out = [[] for _ in range(B)]
  everyone = torch.ones(B, dtype=torch.bool)
  obs = values()
  u.start(obs, everyone)
  o = u.read(obs)['x'].numpy()
  for e in range(B):
    out[e].append(o[e])
  for c in range(1, C):
    first = torch.tensor([c in RESTARTS.get(e, ()) for e in range(B)])
    running = ~first
    if u.needs_substeps:
      for k in range(n):
        world.t += 1
        u.advance(1, running)
        if k < n - 1:
          u.sample(values(), running)
    else:
      world.t += n
      u.advance(n, running)
    obs = values()
    u.sample(obs, running)
    u.start(obs, first)
    o = u.read(obs)['x'].numpy()
    for e in range(B):
      out[e].append(o[e])
  return out, u


GRID = [dict(update_interval=U, delay=D, buffer_size=S, aggregator=None)
        for U, D, S in itertools.product((1, 2, 3, 5, 10), (0, 1, 3, 7), (1, 2, 4))]


@pytest.mark.parametrize('n', [1, 2, 5])
@pytest.mark.parametrize('pad_initial', [False, True])
def test_buffers_equal_the_reference_updater_over_a_grid(ref, n, pad_initial):
  B, C = 4, 16
  for opts in GRID:
    for strip in (False, True):
      want = _run_reference(ref, B, n, C, opts, pad_initial, strip)
      got, u = _run_batched(B, n, C, opts, pad_initial, strip)
      for e in range(B):
        for c in range(C):
          assert got[e][c].shape == want[e][c].shape, (opts, n, strip, e, c, got[e][c].shape, want[e][c].shape)
          assert np.array_equal(got[e][c], want[e][c]), (opts, n, pad_initial, strip, e, c, got[e][c], want[e][c])
      # one launch per control step is kept exactly when every sample falls on a control-step boundary
      assert u.needs_substeps == (opts['update_interval'] % n != 0 and (opts['update_interval'], opts['buffer_size'], opts['delay']) != (1, 1, 0))


@pytest.mark.parametrize('agg', ['min', 'max', 'mean', 'median', 'sum'])
def test_aggregators_equal_numpy_reducers_of_the_reference(ref, agg):
  B, C = 4, 14
  for n, U, D, S in [(2, 2, 0, 3), (5, 1, 2, 4), (3, 3, 3, 2), (2, 1, 0, 2)]:
    opts = dict(update_interval=U, delay=D, buffer_size=S, aggregator=agg)
    want = _run_reference(ref, B, n, C, opts, False, False)
    got, _ = _run_batched(B, n, C, opts, False, False)
    for e in range(B):
      for c in range(C):
        np.testing.assert_allclose(got[e][c], want[e][c], rtol=0, atol=1e-12, err_msg=str((opts, n, e, c)))


def test_options_are_validated_like_the_reference():
  with pytest.raises(KeyError, match='Unrecognized aggregator'):
    bu.Updater(torch, 2, 1, {'x': dict(aggregator='mode')})
  with pytest.raises(ValueError, match='should not be negative'):
    bu.Updater(torch, 2, 1, {'x': dict(delay=-1)})
  with pytest.raises(NotImplementedError, match='constant integer'):
    bu.Updater(torch, 2, 1, {'x': dict(delay=lambda random_state=None: 3)})
  u = bu.Updater(torch, 2, 1, {'y': dict(buffer_size=2)})
  with pytest.raises(KeyError, match='unknown observable'):
    u.start({'x': torch.zeros(2, 3)}, torch.ones(2, dtype=torch.bool))


def _go_to_target(B, **kw):
  from composer_fake import OracleDevicePhysics
  from dm_control_amd.composer import environment
  from dm_control_amd.composer.tasks import go_to_target
  task = go_to_target.GoToTarget()
  phys = OracleDevicePhysics(task.model, B, outputs=('sensordata', 'xpos', 'xmat', 'contact_geom1'))
  return environment.Environment(task, phys, time_limit=0.2, random_state=3, **kw), task


def test_environment_history_of_control_steps_is_one_launch_per_step_and_shifts_by_one():
  """update_interval = the control step, buffer of 3: row i of the buffered observation at control step t is the plain
  observation of control step t - 2 + i; the physics steps of a control step stay one launch."""
  env, task = _go_to_target(2)
  n = env.n_sub_steps
  env_b, _ = _go_to_target(2, observation_options={'joints_pos': dict(update_interval=n, buffer_size=3),
                                                   'sensors_touch': dict(update_interval=n, delay=n)})
  rs = np.random.RandomState(0)
  plain, buffered = [], []
  ts, tb = env.reset(), env_b.reset()
  names = list(ts.observation)
  assert 'joints_pos' in names and 'sensors_touch' in names, names
  for t in range(9):      # time_limit 0.2 s = 6 control steps of 0.03 s: the run crosses an episode boundary
    plain.append(ts), buffered.append(tb)
    a = torch.from_numpy(rs.uniform(-1, 1, (2, task.model.nu)))
    ts, tb = env.step(a), env_b.step(a)
  assert env_b.launches == env.launches      # still one launch per control step
  age = 0
  for t in range(9):
    if int(plain[t].step_type[0]) == 0:
      age = 0
    hist = buffered[t].observation['joints_pos']      # (B, 3, nq - 7)
    assert hist.shape[1] == 3
    for i in range(3):
      back = 2 - i
      want = plain[t - back].observation['joints_pos'] if back <= age else torch.zeros_like(hist[:, i])
      assert torch.equal(hist[:, i], want), (t, i)
    touch = buffered[t].observation['sensors_touch']
    want = plain[t - 1].observation['sensors_touch'] if age >= 1 else torch.zeros_like(touch)
    assert torch.equal(touch, want), t
    for name in names:      # observables without options are untouched
      if name not in ('joints_pos', 'sensors_touch'):
        assert torch.equal(buffered[t].observation[name], plain[t].observation[name]), (t, name)
    age += 1


def test_environment_history_of_physics_steps_takes_one_launch_per_substep():
  """update_interval 1 with a buffer of 3: the last three PHYSICS steps of the control step -- the environment then steps
  launch by launch with an observation pass in between (composer/environment.py:455-458), and the rows are what an
  environment whose control step is one physics step sees under the same held action."""
  from dm_control_amd.composer import environment
  env_b, task = _go_to_target(2, observation_options={'joints_pos': dict(buffer_size=3)})
  n = env_b.n_sub_steps
  env_1, _ = _go_to_target(2, n_sub_steps=1)
  rs = np.random.RandomState(1)
  tb, t1 = env_b.reset(), env_1.reset()
  assert tb.observation['joints_pos'].shape == (2, 3, 56)
  assert torch.equal(tb.observation['joints_pos'][:, 2], t1.observation['joints_pos']) and not tb.observation['joints_pos'][:, :2].any()
  for t in range(3):
    a = torch.from_numpy(rs.uniform(-1, 1, (2, task.model.nu)))
    before = env_b.launches
    tb = env_b.step(a)
    assert env_b.launches - before == n
    fine = [env_1.step(a).observation['joints_pos'] for _ in range(n)]
    for i in range(3):
      # (not bit-equal: the fine environment closes EVERY physics step with the observation-time mj_forward, whose solve
      # leaves another warm start for the next step)
      np.testing.assert_allclose(tb.observation['joints_pos'][:, i].numpy(), fine[n - 3 + i].numpy(), rtol=0, atol=1e-8, err_msg=str((t, i)))
    assert not torch.equal(tb.observation['joints_pos'][:, 0], tb.observation['joints_pos'][:, 2])
  assert int(tb.step_type[0]) == environment.MID


def test_corruptor_acts_on_samples_before_they_are_buffered(ref):
  """observable/base.py:129-138: the aggregator sees corrupted samples; an observable without a ring is corrupted at the read."""
  upd_mod, obs_mod = ref
  world = _World(2)
  corrupt_np = lambda v, random_state=None: np.round(v) * 2.0
  corrupt_t = lambda v: torch.round(v) * 2.0
  for opts in (dict(update_interval=1, delay=1, buffer_size=3, aggregator='sum'), dict(update_interval=1, delay=0, buffer_size=1, aggregator=None)):
    world.t[:] = 0
    obs = [obs_mod.Generic(lambda physics, e=e: world.value(e), corruptor=corrupt_np, **opts) for e in range(2)]
    ups = []
    for o in obs:
      o.enabled = True
      u = upd_mod.Updater({'x': o}, physics_steps_per_control_step=2, strip_singleton_buffer_dim=True)
      u.reset(_NoPhysics(), None)
      ups.append(u)
    b = bu.Updater(torch, 2, 2, {'x': dict(corruptor=corrupt_t, **opts)}, strip_singleton_buffer_dim=True)
    values = lambda: {'x': torch.from_numpy(np.stack([world.value(e) for e in range(2)]))}
    everyone = torch.ones(2, dtype=torch.bool)
    first = b.corrupt(values()); b.start(first, everyone)
    for c in range(6):
      for u in ups:
        u.prepare_for_next_control_step()
      for k in range(2):
        world.t += 1
        for u in ups:
          u.update()
        b.advance(1, everyone)
        if k == 0 and b.needs_substeps:
          b.sample(b.corrupt(values()), everyone)
      o = b.corrupt(values()); b.sample(o, everyone)
      got = b.read(o)['x'].numpy()
      for e in range(2):
        np.testing.assert_array_equal(got[e], np.array(ups[e].get_observation()['x']))
