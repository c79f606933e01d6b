"""The reference's OWN composer / locomotion / soccer unit tests, executed unmodified from /root/reference
(tests/reference_tests.py) on the engine seam of tests/reference_pymjcf.py: `dm_control.mujoco.Physics` is this package's
facade, and the reference's mjcf/physics.py (bindings), composer/environment.py, entity.py, observables, initialisers,
walkers, arenas and tasks run over it exactly as they are on disk (SURVEY 8(a) rows a3 / a11, config 4 and config 5's
host stacks).  CPU tier: the fp64 oracle stands in for the device; `hip` variants (`-m gpu`) step through
`libdmc_hip.so` and need the reference tree on the GPU box (scripts/stage_reference.sh).

A test that ends in `physics.render(...)` (camera observables) is outside this backend's scope: those are counted -- the
expected number per file is pinned below -- and everything else in the file has to pass.  Files NOT run, and why:
mjcf/physics_test.py (its arm model has geom-distance sensors between bodies that carry boxes), locomotion/tasks/
reach_test.py and walkers/rodent_test.py (every test but one builds the rodent's egocentric camera observable; the
rodent's 74 dofs are also beyond the device's nv <= 64), escape / bowl (height fields), the mocap-initialised and
soccer-humanoid walkers (h5py / mocap data absent)."""
import sys

import pytest

import reference_pymjcf as rp
import reference_tests

pytestmark = pytest.mark.skipif(not rp.available(), reason='reference tree not present')

_RENDER = 'rendering is not part of the MI355X physics backend'

# (reference test file, tests it must run, tests that end in physics.render, steps the device)
CASES = [
    ('composer/environment_test.py', 8, 0, True),
    ('composer/environment_hooks_test.py', 1, 0, True),
    ('composer/entity_test.py', 245, 0, True),
    ('composer/initializers/prop_initializer_test.py', 9, 0, True),
    ('composer/observation/updater_test.py', 7, 0, False),
    ('composer/observation/obs_buffer_test.py', 5, 0, False),
    ('composer/observation/observable/base_test.py', 6, 1, True),
    ('composer/observation/observable/mjcf_test.py', 8, 2, True),
    ('entities/props/position_detector_test.py', 4, 0, True),
    ('entities/props/primitive_test.py', 13, 0, True),
    ('locomotion/tasks/go_to_target_test.py', 5, 0, True),
    ('locomotion/tasks/corridors_test.py', 3, 1, True),
    ('locomotion/soccer/task_test.py', 35, 0, True),
    ('locomotion/soccer/pitch_test.py', 6, 0, True),
    ('locomotion/soccer/boxhead_test.py', 6, 0, True),
    ('locomotion/soccer/soccer_ball_test.py', 3, 0, True),
    ('locomotion/walkers/cmu_humanoid_test.py', 27, 1, True),
    ('locomotion/walkers/ant_test.py', 10, 3, True),
    ('locomotion/walkers/jumping_ball_test.py', 17, 3, True),
    ('locomotion/walkers/base_test.py', 2, 0, False),
    ('locomotion/walkers/rescale_test.py', 1, 0, False),
    ('locomotion/walkers/scaled_actuators_test.py', 4, 0, True),
    ('locomotion/arenas/floors_test.py', 3, 0, True),
    ('locomotion/arenas/corridors_test.py', 6, 0, True),
    ('locomotion/props/target_sphere_test.py', 1, 0, True),
]
# soccer/task_test.py's TaskTest.test_render builds a tracking camera (engine.MovableCamera): render-only, not loaded
SKIP = {'locomotion/soccer/task_test.py': ('TaskTest.test_render',)}


@pytest.fixture
def engine():
  rp.bind_engine()
  rp._soccer_modules()      # pylint: disable=protected-access  (the soccer package without its h5py-bound walkers)
  yield sys.modules['dm_control.composer']
  rp.unload()


def _run(path, ntests, nrender):
  result, report = reference_tests.run(path, None, skip=SKIP.get(path, ()))
  assert result.testsRun >= ntests, (result.testsRun, report)
  render = [t for t, tb in result.errors if _RENDER in tb]
  others = [(str(t), tb) for t, tb in result.errors if _RENDER not in tb]
  assert not result.failures and not others, report
  assert len(render) == nrender, [str(t) for t in render]


@pytest.mark.parametrize('path,ntests,nrender', [c[:3] for c in CASES])
def test_reference_unit_test_file_passes_on_the_facade(engine, oracle_backend, path, ntests, nrender):
  _run(path, ntests, nrender)


@pytest.mark.gpu
@pytest.mark.parametrize('path,ntests,nrender', [c[:3] for c in CASES if c[3]])
def test_reference_unit_test_file_passes_on_the_hip_path(engine, path, ntests, nrender):
  _run(path, ntests, nrender)
