"""PGS (north_star: "PGS/Newton constraint solve"; option solver="PGS", dm_control/mjcf/schema.xml:69-72).  The
reference's own hot-path test asset runs it (mujoco/testing/assets/humanoid.xml:9: PGS + RK4, used by
mujoco/wrapper/core_test.py and mujoco/engine_test.py); suite/assets/testing_humanoid_pgs.xml is its physics-only
restatement (byte-identical compiled blob).

CPU tier: (1) the oracle's sweep against an independent numpy Gauss-Seidel written from MuJoCo's published algorithm,
(2) its fixed point against scipy's bounded minimiser of the dual cost and against the Newton solver's optimum,
(3) the kernel core (tests/emu) against the oracle, pyramidal and elliptic (condim 3 / 4 / 6)."""
import os

import numpy as np
import pytest

from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.suite import common
from emu_lib import EmuPhysics
from oracle.oracle import OraclePhysics

from ref_root import REF  # noqa: E402
REF_XML = REF + '/mujoco/testing/assets/humanoid.xml'
CT_FRICTION_DOF, CT_ELLIPTIC, CT_EQUALITY = 4, 3, 6


def _xml(cone='pyramidal', floor_condim=3, iterations=None, tolerance=None, solver='PGS', integrator=None):
  x = common.read_model('testing_humanoid_pgs.xml')
  opt = 'solver="%s"' % solver
  if cone == 'elliptic':
    opt += ' cone="elliptic"'
  if tolerance is not None:
    opt += ' tolerance="%g"' % tolerance
  x = x.replace('solver="PGS"', opt)
  if iterations is not None:
    x = x.replace('iterations="50"', 'iterations="%d"' % iterations)
  if integrator is not None:
    x = x.replace('integrator="RK4"', 'integrator="%s"' % integrator)
  if floor_condim != 3:
    assert 'condim="3" friction="1 .1 .1"' in x
    x = x.replace('condim="3" friction="1 .1 .1"', 'condim="%d" friction="1 .1 .1"' % floor_condim)
  return x


def _fallen_state(m, seed, steps=450):
  """A state with the humanoid on the floor (several frictional contacts + joint limits), reached with Newton."""
  mn = mc.compile_xml(_xml(solver='Newton', integrator='Euler'))
  p = OraclePhysics(mn)
  rs = np.random.RandomState(seed)
  q = mn.qpos0.copy(); q[7:] += rs.uniform(-.4, .4, mn.nq - 7)
  p.qpos[:] = q
  p.forward()
  for _ in range(steps):
    p.set_control(rs.uniform(-1, 1, mn.nu))
    p.step()
  return p.qpos.copy(), p.qvel.copy(), p.qacc_warmstart.copy(), rs.uniform(-1, 1, mn.nu)


@pytest.mark.skipif(not os.path.exists(REF_XML), reason='reference tree not present')
def test_restated_asset_compiles_to_the_reference_blob():
  a = mc.compile_xml(open(REF_XML).read())
  b = mc.compile_xml(common.read_model('testing_humanoid_pgs.xml'))
  assert a.opt.solver == 0 and a.opt.integrator == 1 and a.opt.iterations == 50      # PGS, RK4
  for x, y in zip(a.pack(), b.pack()):
    np.testing.assert_array_equal(x, y)


def _dual_problem(m, p):
  """AR = J M^-1 J' + diag(R), b = J qacc_smooth - aref, from the oracle's position / velocity stage."""
  nv, nefc = m.nv, p.nefc
  J = np.array(p.efc_J[:nefc * nv]).reshape(nefc, nv)
  M = np.array(p.qM).reshape(nv, nv)
  AR = J @ np.linalg.solve(M, J.T) + np.diag(np.array(p.efc_R[:nefc]))
  b = J @ np.array(p.qacc_smooth) - np.array(p.efc_aref[:nefc])
  return J, M, AR, b


def _numpy_pgs_pyramidal(m, p, AR, b, f0, sweeps):
  """mj_solPGS for scalar rows, written independently of the oracle: unconstrained minimum of the row, clamp
  (equality: none, dof friction: +-frictionloss, everything else: >= 0), undo if the cost went up by > 1e-10."""
  f = f0.copy()
  types, ids = p.efc_type, p.efc_id
  for _ in range(sweeps):
    for i in range(len(b)):
      res = b[i] + AR[i] @ f
      old = f[i]
      new = old - res / AR[i, i]
      if types[i] == CT_FRICTION_DOF:
        fl = m.dof_frictionloss[ids[i]]
        new = min(max(new, -fl), fl)
      elif types[i] != CT_EQUALITY:
        new = max(new, 0.0)
      d = new - old
      if 0.5 * d * d * AR[i, i] + d * res > 1e-10:
        new = old
      f[i] = new
  return f


@pytest.mark.parametrize('sweeps', [1, 2, 5])
def test_oracle_sweeps_equal_an_independent_numpy_gauss_seidel(sweeps):
  m = mc.compile_xml(_xml(iterations=sweeps, tolerance=0.0, integrator='Euler'))
  q, v, w, a = _fallen_state(m, 6)
  p = OraclePhysics(m)
  p.qpos[:] = q; p.qvel[:] = v; p.set_control(a)
  p.model.opt_int('disableflags', p.model.opt_int('disableflags') | (1 << 9))      # mjDSBL_WARMSTART: start from f = 0
  p.forward()
  assert p.nefc >= 12 and p.ncon >= 3 and p.solver_iter == sweeps
  J, M, AR, b = _dual_problem(m, p)
  want = _numpy_pgs_pyramidal(m, p, AR, b, np.zeros(p.nefc), sweeps)
  np.testing.assert_allclose(np.array(p.efc_force[:p.nefc]), want, rtol=1e-9, atol=1e-9)
  # dualFinish: qacc = qacc_smooth + M^-1 J' f
  np.testing.assert_allclose(np.array(p.qacc), np.array(p.qacc_smooth) + np.linalg.solve(M, J.T @ want), rtol=1e-9, atol=1e-8)


def test_oracle_fixed_point_is_the_dual_optimum_and_agrees_with_newton():
  from scipy import optimize
  m = mc.compile_xml(_xml(iterations=20000, tolerance=1e-15, integrator='Euler'))
  q, v, w, a = _fallen_state(m, 6)
  p = OraclePhysics(m)
  p.qpos[:] = q; p.qvel[:] = v; p.qacc_warmstart[:] = w; p.set_control(a)
  p.forward()
  nefc = p.nefc
  assert nefc >= 12
  J, M, AR, b = _dual_problem(m, p)
  f = np.array(p.efc_force[:nefc])
  lo = np.zeros(nefc); hi = np.full(nefc, np.inf)
  for i in range(nefc):
    if p.efc_type[i] == CT_EQUALITY:
      lo[i] = -np.inf
    elif p.efc_type[i] == CT_FRICTION_DOF:
      lo[i], hi[i] = -m.dof_frictionloss[p.efc_id[i]], m.dof_frictionloss[p.efc_id[i]]
  assert np.all(f >= lo - 1e-12) and np.all(f <= hi + 1e-12)
  cost = lambda x: 0.5 * x @ AR @ x + x @ b
  r = optimize.minimize(cost, f, jac=lambda x: AR @ x + b, bounds=list(zip(lo, hi)), method='L-BFGS-B',
                        options=dict(ftol=1e-16, gtol=1e-12, maxiter=20000))
  assert cost(f) <= r.fun + 1e-9 * max(1.0, abs(r.fun))          # the minimiser cannot improve on it
  g = AR @ f + b                                                  # KKT: zero gradient on free rows, >= 0 at the lower bound
  free = (f > lo + 1e-9) & (f < hi - 1e-9)
  assert np.abs(g[free]).max() < 1e-6 * max(1.0, np.abs(b).max())
  assert g[(f <= lo + 1e-9) & np.isfinite(lo)].min() > -1e-6 * max(1.0, np.abs(b).max())
  # primal / dual agreement: the Newton solver's qacc at the same state
  mn = mc.compile_xml(_xml(solver='Newton', integrator='Euler', tolerance=1e-15))
  pn = OraclePhysics(mn)
  pn.qpos[:] = q; pn.qvel[:] = v; pn.qacc_warmstart[:] = w; pn.set_control(a)
  pn.forward()
  np.testing.assert_allclose(np.array(p.qacc), np.array(pn.qacc), rtol=1e-6, atol=1e-6 * np.abs(pn.qacc).max())


@pytest.mark.parametrize('cone,condim', [('pyramidal', 3), ('elliptic', 3), ('elliptic', 4), ('elliptic', 6), ('pyramidal', 4)])
def test_kernel_core_matches_oracle_per_step(cone, condim):
  """Teacher-forced (state AND warm start from the oracle before every step): PGS stops at its iteration cap far from
  convergence, so its result depends on the warm start and open-loop trajectories separate by chaos, not by logic."""
  m = mc.compile_xml(_xml(cone=cone, floor_condim=condim))
  p = OraclePhysics(m)
  g = EmuPhysics(m, prec=64)
  rs = np.random.RandomState(1)
  q = m.qpos0.copy(); q[2] = 0.9; q[7:] += rs.uniform(-.3, .3, m.nq - 7)
  p.qpos[:] = q
  p.forward()
  worst, max_nefc, max_iter = 0.0, 0, 0
  for t in range(160):
    a = rs.uniform(-1, 1, m.nu)
    g.qpos[:] = p.qpos; g.qvel[:] = p.qvel; g.qacc_warmstart[:] = p.qacc_warmstart
    p.set_control(a); g.ctrl[:] = a
    p.step(); g.step()
    worst = max(worst, np.abs(p.qpos - g.qpos).max(), np.abs(p.qvel - g.qvel).max() * 1e-2)
    max_nefc, max_iter = max(max_nefc, p.nefc), max(max_iter, p.solver_iter)
    assert g.solver_iter[0] == p.solver_iter, t
  assert worst < 1e-10, worst
  assert max_nefc >= 12 and max_iter >= 10
  assert not g.warning.any()


def test_kernel_core_fp32_one_step_error():
  m = mc.compile_xml(_xml())
  p = OraclePhysics(m)
  g = EmuPhysics(m, prec=32)
  rs = np.random.RandomState(2)
  q = m.qpos0.copy(); q[2] = 0.9; q[7:] += rs.uniform(-.3, .3, m.nq - 7)
  p.qpos[:] = q
  p.forward()
  errs = []
  for t in range(160):
    a = rs.uniform(-1, 1, m.nu).astype(np.float32).astype(np.float64)
    g.qpos[:] = p.qpos; g.qvel[:] = p.qvel; g.qacc_warmstart[:] = p.qacc_warmstart
    p.set_control(a); g.ctrl[:] = a
    p.step(); g.step()
    errs.append(np.abs(p.qpos - g.qpos).max() / max(1.0, np.abs(p.qpos).max()))
  assert np.median(errs) < 1e-6 and max(errs) < 1e-4, (np.median(errs), max(errs))


def test_pgs_with_noslip_and_open_loop_short_horizon():
  x = _xml(cone='elliptic').replace('solver="PGS"', 'solver="PGS" noslip_iterations="3"')
  m = mc.compile_xml(x)
  p = OraclePhysics(m)
  g = EmuPhysics(m, prec=64)
  rs = np.random.RandomState(4)
  q = m.qpos0.copy(); q[2] = 0.9; q[7:] += rs.uniform(-.3, .3, m.nq - 7)
  p.qpos[:] = q; g.qpos[:] = q
  p.forward()
  for t in range(60):
    a = rs.uniform(-1, 1, m.nu)
    p.set_control(a); g.ctrl[:] = a
    p.step(); g.step()
  assert p.ncon > 0
  np.testing.assert_allclose(g.qpos, p.qpos, rtol=0, atol=1e-7)
