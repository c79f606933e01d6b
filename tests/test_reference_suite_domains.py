"""The reference's OWN suite domains -- `dm_control/suite/<domain>.py` executed unmodified from /root/reference, with
the reference's own XML and its own `control.Environment` -- drive this package's Physics facade (CPU tier: the fp64
oracle stands in for the device), and every episode equals the one `dm_control_amd.suite.load(domain, task)` produces
with this package's task port and its physics-only restatement of the XML: step types, rewards, discounts and every
observation, element for element.  What it pins: the task ports (initialisation incl. the order random numbers are
drawn in, rewards, observations) and the restated assets.  Skips where the
reference tree is absent (the GPU box)."""
import numpy as np
import pytest

import reference_loader

pytestmark = pytest.mark.skipif(not reference_loader.available(), reason='reference tree not present')

# (domain, task, steps): every domain of the reference suite but the dog (meshes, 150 dofs) and quadruped escape (hfield)
CASES = [('acrobot', 'swingup', 12), ('acrobot', 'swingup_sparse', 12),
         ('ball_in_cup', 'catch', 12),
         ('cheetah', 'run', 12),
         ('finger', 'spin', 12), ('finger', 'turn_easy', 12), ('finger', 'turn_hard', 12),
         ('fish', 'upright', 12), ('fish', 'swim', 12),
         ('hopper', 'stand', 12), ('hopper', 'hop', 12),
         ('humanoid', 'stand', 8), ('humanoid', 'walk', 8), ('humanoid', 'run', 8), ('humanoid', 'run_pure_state', 8),
         ('humanoid_CMU', 'stand', 4), ('humanoid_CMU', 'run', 4),
         ('pendulum', 'swingup', 12),
         ('point_mass', 'easy', 12), ('point_mass', 'hard', 12),
         ('reacher', 'easy', 12), ('reacher', 'hard', 12),
         ('walker', 'stand', 12), ('walker', 'walk', 12), ('walker', 'run', 12),
         # modules that build their XML with lxml (an xml.etree adapter stands in for it here)
         ('cartpole', 'balance', 12), ('cartpole', 'balance_sparse', 12), ('cartpole', 'swingup', 12),
         ('cartpole', 'swingup_sparse', 12), ('cartpole', 'two_poles', 12), ('cartpole', 'three_poles', 12),
         ('swimmer', 'swimmer6', 8), ('swimmer', 'swimmer15', 6),
         ('lqr', 'lqr_2_1', 12), ('lqr', 'lqr_6_2', 12),
         ('manipulator', 'bring_ball', 8), ('manipulator', 'bring_peg', 8), ('manipulator', 'insert_ball', 8),
         ('manipulator', 'insert_peg', 8),
         ('stacker', 'stack_2', 8), ('stacker', 'stack_4', 8),
         ('quadruped', 'walk', 6), ('quadruped', 'run', 6), ('quadruped', 'fetch', 6)]


@pytest.fixture
def ref_suite():
  yield reference_loader
  reference_loader.unload()


def _episode(env, actions):
  ts = env.reset()
  out = [ts]
  for a in actions:
    out.append(env.step(a))
  return out


# a second seed for the domains whose initialisation loops until a sample is accepted (contact-free poses) or draws a
# data-dependent number of values
CASES_2 = [(d, t, n, 4) for d, t, n in CASES if d in ('humanoid', 'humanoid_CMU', 'manipulator', 'stacker', 'quadruped', 'finger')]


@pytest.mark.parametrize('domain,task,nsteps,seed', [c + (11,) for c in CASES] + CASES_2)
def test_reference_domain_module_unmodified_equals_the_task_port(ref_suite, oracle_backend, domain, task, nsteps, seed):
  _check_domain(ref_suite, domain, task, nsteps, seed)


# the same comparison with NOTHING standing in for the device: the reference's domain module, Task and control.Environment
# step `libdmc_hip.so` through the facade (needs the reference tree on the GPU box: scripts/stage_reference.sh)
GPU_CASES = [('cheetah', 'run', 12), ('cartpole', 'swingup', 12), ('humanoid', 'walk', 8), ('humanoid_CMU', 'run', 4),
             ('walker', 'walk', 12), ('hopper', 'hop', 12), ('finger', 'turn_hard', 12), ('fish', 'swim', 12),
             ('manipulator', 'bring_ball', 8), ('stacker', 'stack_4', 8), ('quadruped', 'fetch', 6), ('swimmer', 'swimmer6', 8),
             ('reacher', 'hard', 12), ('ball_in_cup', 'catch', 12), ('lqr', 'lqr_6_2', 12), ('acrobot', 'swingup', 12),
             ('pendulum', 'swingup', 12), ('point_mass', 'hard', 12)]


@pytest.mark.gpu
@pytest.mark.parametrize('domain,task,nsteps', GPU_CASES)
def test_reference_domain_module_unmodified_on_the_hip_path(ref_suite, domain, task, nsteps):
  # the reference module builds its Physics with the facade's automatic contact capacity, suite.load with the tuned one
  # of suite/common.py: another kernel instantiation (model-specialised or generic), so rounding-level differences
  _check_domain(ref_suite, domain, task, nsteps, 11, atol=1e-9)


def _check_domain(ref_suite, domain, task, nsteps, seed, atol=0):
  from dm_control_amd import suite
  mod = ref_suite.load(domain)
  assert mod.__file__.startswith(reference_loader.REF) and task in mod.SUITE
  ref_env = mod.SUITE[task](random=seed)
  ours = suite.load(domain, task, task_kwargs=dict(random=seed))
  assert type(ref_env).__module__ == 'dm_control.rl.control' and type(ref_env.task).__module__ == 'dm_control.suite.' + domain
  spec, ospec = ref_env.action_spec(), ours.action_spec()
  assert spec.shape == ospec.shape
  np.testing.assert_array_equal(spec.minimum, ospec.minimum); np.testing.assert_array_equal(spec.maximum, ospec.maximum)
  assert ref_env.control_timestep() == ours.control_timestep()
  assert ref_env._step_limit == ours._step_limit and ref_env._n_sub_steps == ours._n_sub_steps
  lo = np.where(np.isfinite(spec.minimum), spec.minimum, -1.0); hi = np.where(np.isfinite(spec.maximum), spec.maximum, 1.0)
  acts = np.random.RandomState(5).uniform(lo, hi, (nsteps,) + spec.shape)
  a, b = _episode(ref_env, acts), _episode(ours, acts)
  for t, (x, y) in enumerate(zip(a, b)):
    assert int(x.step_type) == int(y.step_type), t
    assert x.discount == y.discount and (x.reward == y.reward or abs(x.reward - y.reward) <= atol), (t, x.reward, y.reward)
    assert list(x.observation) == list(y.observation), t
    for k in x.observation:
      np.testing.assert_allclose(np.asarray(x.observation[k]), np.asarray(y.observation[k]), rtol=0, atol=atol, err_msg='%s step %d' % (k, t))
  # the observation specs the two environments advertise agree as well
  rs, os_ = ref_env.observation_spec(), ours.observation_spec()
  assert list(rs) == list(os_) and all(rs[k].shape == os_[k].shape and rs[k].dtype == os_[k].dtype for k in rs)
