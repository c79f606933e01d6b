"""The reference's own unit tests, executed unmodified (tests/reference_tests.py) against this package's modules."""
import pytest

import reference_tests

pytestmark = pytest.mark.skipif(not reference_tests.available(), reason='reference tree not present')


def _check(result, report, min_tests):
  assert result.testsRun >= min_tests, (result.testsRun, report)
  assert result.wasSuccessful(), report


def test_reference_rl_control_test_passes_on_envs_control():
  """dm_control/rl/control_test.py (Environment call order, time limits, flat observations, compute_n_steps) against
  dm_control_amd.envs.control -- the module suite / composer-free tasks are driven by."""
  from dm_control_amd.envs import control
  result, report = reference_tests.run('rl/control_test.py', {'dm_control.rl.control': control})
  _check(result, report, 12)


def test_reference_rewards_test_passes_on_suite_rewards():
  """dm_control/utils/rewards_test.py (tolerance(): every sigmoid, margins, bounds, vectorised inputs, errors)."""
  from dm_control_amd.suite import rewards
  result, report = reference_tests.run('utils/rewards_test.py', {'dm_control.utils.rewards': rewards})
  _check(result, report, 10)


def _suite_modules():
  import types
  from dm_control_amd import suite
  from dm_control_amd.envs import control
  constants = types.ModuleType('dm_control.mujoco.wrapper.mjbindings.constants')
  constants.mjMAXVAL = 1e10      # mjMAXVAL (the bound check_invalid_state uses; tasks clip action bounds against it)
  return {'dm_control.suite': suite, 'dm_control.rl.control': control,
          'dm_control.mujoco.wrapper.mjbindings.constants': constants}


def test_reference_suite_loader_test_passes_on_this_suite(oracle_backend):
  result, report = reference_tests.run('suite/loader_test.py', _suite_modules())
  _check(result, report, 3)


# suite/suite_test.py: 9 checks x every registered task.  Out of this backend's scope and skipped: the two that look at
# rendering content of the XML (at least two cameras per model; reward-coloured materials) -- the assets here are
# physics-only restatements.
_SUITE_SKIP = ('SuiteTest.test_model_has_at_least_2_cameras', 'SuiteTest.test_visualize_reward')
_SUITE_GROUPS = ('test_constants', 'test_components_have_names', 'test_task_conforms_to_spec',
                 'test_environment_is_deterministic', 'test_task_supports_environment_kwargs',
                 'test_observation_arrays_dont_share_memory', 'test_observations_dont_contain_constant_elements',
                 'test_initial_state_is_randomized')


@pytest.mark.parametrize('group', _SUITE_GROUPS)
def test_reference_suite_test_passes_on_this_suite(oracle_backend, group):
  """dm_control/suite/suite_test.py unmodified, all 45 tasks of dm_control_amd.suite: names, specs, reward / discount
  ranges of the benchmarking tasks, determinism under a seed, environment kwargs, no aliasing between consecutive
  observations, no constant observation entries over two 1000-step episodes, randomised initial states."""
  skip = _SUITE_SKIP + tuple('SuiteTest.' + g for g in _SUITE_GROUPS if g != group)
  result, report = reference_tests.run('suite/suite_test.py', _suite_modules(), skip=skip)
  _check(result, report, 1)


def _exec_reference_module(name, rel):
  import importlib.util
  import sys
  spec = importlib.util.spec_from_file_location(name, reference_tests.REF + '/' + rel)
  mod = importlib.util.module_from_spec(spec)
  sys.modules[name] = mod
  spec.loader.exec_module(mod)
  return mod


@pytest.mark.parametrize('wrapper', ['action_scale', 'action_noise'])
def test_reference_action_wrappers_and_their_tests_on_this_packages_environment(oracle_backend, wrapper):
  """suite/wrappers/action_scale.py / action_noise.py are pure dm_env wrappers: executed unmodified over this package's
  dm_env shim they (a) pass their own unit tests with `dm_control.rl.control` = envs/control.py and (b) wrap an actual
  `suite.load` environment of this package."""
  import sys
  import numpy as np
  from dm_control_amd import suite
  from dm_control_amd.envs import control, dm_env_api
  saved = {k: sys.modules.get(k) for k in ('dm_env', 'dm_env.specs')}
  sys.modules['dm_env'], sys.modules['dm_env.specs'] = dm_env_api, dm_env_api.specs
  try:
    mod = _exec_reference_module('dm_control.suite.wrappers.' + wrapper, 'suite/wrappers/%s.py' % wrapper)
    result, report = reference_tests.run('suite/wrappers/%s_test.py' % wrapper,
                                         {'dm_control.rl.control': control, 'dm_control.suite.wrappers.' + wrapper: mod})
    _check(result, report, 4)
    env = suite.load('cheetah', 'run', task_kwargs=dict(random=0))
    if wrapper == 'action_scale':
      wrapped = mod.Wrapper(env, minimum=-5., maximum=5.)
      spec = wrapped.action_spec()
      assert spec.minimum.min() == -5 and spec.maximum.max() == 5 and spec.shape == (6,)
      wrapped.reset()
      wrapped.step(np.full(6, 5.0))      # +5 maps to the cheetah's +1
      np.testing.assert_allclose(env.physics.data.ctrl, np.ones(6))
    else:
      wrapped = mod.Wrapper(env, scale=0.1)      # (its noise comes from the TASK's RandomState)
      wrapped.reset()
      ts = wrapped.step(np.zeros(6))
      ctrl = np.array(env.physics.data.ctrl)
      assert ts.reward is not None and 0 < np.abs(ctrl).max() < 1.0      # noise of 0.1 x the range, clipped into it
  finally:
    sys.modules.pop('dm_control.suite.wrappers.' + wrapper, None)
    for k, v in saved.items():
      if v is None:
        sys.modules.pop(k, None)
      else:
        sys.modules[k] = v


def _with_dm_env(fn):
  import sys
  from dm_control_amd.envs import dm_env_api
  saved = {k: sys.modules.get(k) for k in ('dm_env', 'dm_env.specs')}
  sys.modules['dm_env'], sys.modules['dm_env.specs'] = dm_env_api, dm_env_api.specs
  try:
    return fn()
  finally:
    sys.modules.pop('dm_control.suite.wrappers.mujoco_profiling', None)
    for k, v in saved.items():
      if v is None:
        sys.modules.pop(k, None)
      else:
        sys.modules[k] = v


def _profiling_wrapper_checks():
  import numpy as np
  from dm_control_amd import suite
  mod = _exec_reference_module('dm_control.suite.wrappers.mujoco_profiling', 'suite/wrappers/mujoco_profiling.py')
  result, report = reference_tests.run('suite/wrappers/mujoco_profiling_test.py',
                                       {'dm_control.suite': suite, 'dm_control.suite.cartpole': suite.cartpole,
                                        'dm_control.suite.wrappers.mujoco_profiling': mod})
  _check(result, report, 1)
  # and on a stepping environment: the step timer advances by the physics steps of each control step (cheetah: 1 per
  # control step ... cartpole swingup: 1; humanoid: 5), its duration grows, and stays put across observation reads
  env = suite.load('humanoid', 'stand', task_kwargs=dict(random=0))
  wrapped = mod.Wrapper(env)
  ts = wrapped.reset()
  d0, n0 = ts.observation['step_timing']
  nsub = env._n_sub_steps
  assert nsub == 5
  for k in range(1, 4):
    ts = wrapped.step(np.zeros(env.action_spec().shape))
    d, n = ts.observation['step_timing']
    assert n == n0 + k * nsub and d > d0, (k, d, n)
    d0_prev = d
  assert env.physics.data.timer[0].number == n and env.physics.data.timer[0].duration == d0_prev
  assert env.physics.data.timer[1].number >= 1      # the mj_forward launches of reset (mjTIMER_FORWARD)
  assert env.physics.data.timer[5].number == 0 and len(env.physics.data.timer) == 15


def test_reference_mujoco_profiling_wrapper_and_its_test_on_this_packages_physics(oracle_backend):
  """suite/wrappers/mujoco_profiling.py calls `env.physics.enable_profiling()` (mujoco/engine.py:135-137) and
  reads `physics.data.timer[0].duration / .number` (mujoco_profiling.py:94-103): the wrapper and its own unit test run
  unmodified on this package's Physics (CPU tier: the oracle stand-in times itself with perf_counter)."""
  _with_dm_env(_profiling_wrapper_checks)


@pytest.mark.gpu
def test_reference_mujoco_profiling_wrapper_on_the_hip_path():
  """The same on the device: the timers are hipEvent brackets around every launch (dmc_batch_enable_profiling)."""
  _with_dm_env(_profiling_wrapper_checks)
