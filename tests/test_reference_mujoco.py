"""The reference's OWN `dm_control/mujoco/engine.py`, `wrapper/core.py` and `index.py` -- imported unmodified from
/root/reference (tests/reference_mujoco.py) -- with `mujoco` = dm_control_amd.mujoco_api, and the reference's own unit tests
of exactly those files executed on that stack (the calls on `(model.ptr, data.ptr)`):

  mujoco/engine_test.py          Physics.step / forward / reset / get_state / set_state(sig) / copy / pickle / warnings
  mujoco/wrapper/core_test.py    MjModel / MjData loading, copying, pickling, every attribute read / written / stepped,
                                 callbacks, disable flags, mj_objectVelocity, mj_contactForce KATs
  mujoco/index_test.py           named indexing over mjbindings.sizes
  mujoco/thread_safety_test.py   independent Physics objects loaded and stepped from 4 threads
  suite/lqr_test.py              lqr_solver.solve (mju_sym2dense of data.M) against the measured cost of the rollout

CPU tier: the fp64 oracle stands in for the device (`mujoco_api.BatchedPhysics` swapped for tests/oracle_backend.OracleBatch);
`-m gpu`: the same files step through libdmc_hip.so (the reference tree staged by scripts/stage_reference.sh).

Pinned exclusions, by name: tests that END in rendering (Camera / MjvScene / MjrContext: out of scope) and
the one model that needs a MuJoCo plugin (`mujoco.elasticity.cable`, a <composite>).  Everything else must pass."""
import sys

import pytest

import reference_mujoco as rm
import reference_tests

pytestmark = pytest.mark.skipif(not rm.available(), reason='reference tree not present')

_RENDER = ('rendering is not part of the MI355X physics backend', 'No OpenGL rendering backend is available')

# file -> (tests that must run, tests that end in rendering, names skipped outright)
CASES = {
    'mujoco/engine_test.py': (58, 24, ('MujocoEngineTest.testSetGetPhysicsStateWithPlugin',)),
    'mujoco/wrapper/core_test.py': (300, 2, ()),
    'mujoco/index_test.py': (310, 0, ()),
    'mujoco/thread_safety_test.py': (7, 2, ()),
    'suite/lqr_test.py': (2, 0, ()),
}


@pytest.fixture
def seam():
  rm.load()
  yield sys.modules['dm_control.mujoco']
  rm.unload()


@pytest.fixture
def oracle_device(monkeypatch):
  import oracle_backend as ob
  from dm_control_amd import mujoco_api
  monkeypatch.setattr(mujoco_api, 'BatchedPhysics', ob.OracleBatch)


def _run(path):
  ntests, nrender, skip = CASES[path]
  result, report = reference_tests.run(path, None, skip=skip)
  assert result.testsRun >= ntests, (result.testsRun, report[-3000:])
  render = [t for t, tb in result.errors if any(r in tb for r in _RENDER)]
  others = [(str(t), tb) for t, tb in result.errors if not any(r in tb for r in _RENDER)]
  assert not result.failures and not others, report[-6000:]
  assert len(render) == nrender, sorted(str(t) for t in render)


@pytest.mark.parametrize('path', sorted(CASES))
def test_reference_mujoco_test_file_passes_on_the_seam(oracle_device, seam, path):
  _run(path)


@pytest.mark.gpu
@pytest.mark.parametrize('path', sorted(CASES))
def test_reference_mujoco_test_file_passes_on_the_hip_path(seam, path):
  _run(path)
