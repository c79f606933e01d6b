"""Independent pins of conventions the oracle restates without a MuJoCo binary (PARITY_ASSUMPTIONS.md rows 5-8, 13,
22, 23, 33).  Nothing here shares code with oracle/mjstep_oracle.c or the HIP kernel: each check recomputes the
quantity from its published DEFINITION with numpy / scipy (finite differences, generic minimisers, brute-force
sampling) and compares the oracle against it.

  rows 6 / 7 / 8 / 22 / 23: the constraint solve is, by MuJoCo's documentation (computation/index.html "convex
      optimization"), the unique minimiser of  1/2 (a - a_s)' M (a - a_s) + sum_i s_i(J_i a - aref_i)  with the per-type
      costs s_i.  A generic scipy minimiser of that function, written from the definition, must land on the oracle's
      qacc whatever path Newton took, whether the warm start was kept, and whether islands are solved jointly.
  row 5: body_invweight0 / dof_invweight0 from M^-1 at qpos0, with M built from finite-difference body Jacobians.
  row 13: capsule-capsule closest points against dense sampling of both segments.
  row 33: sphere-box / capsule-box / box-box distances against dense sampling of the surfaces.
"""
import os

import numpy as np
import pytest
from scipy import optimize

from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.suite import common
from oracle.oracle import OraclePhysics

CT_LIMIT, CT_FRICTIONLESS, CT_PYRAMIDAL, CT_ELLIPTIC, CT_FRICTION_DOF, CT_LIMIT_TENDON, CT_EQUALITY = range(7)


# ---- rows 6 / 7 / 8 / 22 / 23: the solver's answer is the minimiser of the documented cost ---------------------
def _constraint_cost(m, o):
  """(cost(a), grad(a)) of MuJoCo's primal problem at the oracle's current constraint set, from the definition."""
  nv, ne = m.nv, o.nefc
  M = np.array(o.qM).reshape(nv, nv)
  J = np.array(o.efc_J)[:ne * nv].reshape(ne, nv)
  aref, D = np.array(o.efc_aref)[:ne], np.array(o.efc_D)[:ne]
  typ, cid = np.array(o.efc_type), np.array(o.efc_id)
  a_s = np.array(o.qacc_smooth)
  floss = np.asarray(m.dof_frictionloss)
  # elliptic blocks: rows of one contact are consecutive, first row = normal
  blocks, i = [], 0
  while i < ne:
    if typ[i] == CT_ELLIPTIC:
      j = i
      while j < ne and typ[j] == CT_ELLIPTIC and cid[j] == cid[i]:
        j += 1
      blocks.append((i, j))
      i = j
    else:
      i += 1
  contacts = {c: o.contact(c) for c in set(cid[typ == CT_ELLIPTIC])}

  def f(a):
    r = J @ a - aref
    cost = 0.5 * (a - a_s) @ M @ (a - a_s)
    g_r = np.zeros(ne)
    for k in range(ne):
      t = typ[k]
      if t == CT_EQUALITY:
        cost += 0.5 * D[k] * r[k]**2; g_r[k] = D[k] * r[k]
      elif t == CT_FRICTION_DOF:      # Huber: quadratic inside |r| < R f, linear outside
        fl = floss[cid[k]]; R = 1.0 / D[k]
        if abs(r[k]) < R * fl:
          cost += 0.5 * D[k] * r[k]**2; g_r[k] = D[k] * r[k]
        else:
          cost += fl * (abs(r[k]) - 0.5 * R * fl); g_r[k] = fl * np.sign(r[k])
      elif t in (CT_LIMIT, CT_LIMIT_TENDON, CT_FRICTIONLESS, CT_PYRAMIDAL):
        if r[k] < 0:
          cost += 0.5 * D[k] * r[k]**2; g_r[k] = D[k] * r[k]
    for (i0, i1) in blocks:
      # elliptic cone (MuJoCo docs, "Elliptic cones"): in the scaled variables u = (mu r_N, fr_k r_Tk) with
      # mu = friction_1 sqrt(D_N / D_T1): top zone N >= mu T -> 0; bottom zone mu N + T <= 0 -> per-row quadratic;
      # middle zone -> 1/2 Dm (N - mu T)^2, Dm = D_N / (mu^2 (1 + mu^2))
      c = contacts[cid[i0]]
      fr = np.asarray(c['friction'])
      dim = i1 - i0
      fj = fr[:dim - 1]                  # contact friction: (tangent 1, tangent 2, spin, roll 1, roll 2)
      mu = fr[0] * np.sqrt(D[i0] / D[i0 + 1])
      rr = r[i0:i1]
      N = rr[0] * mu
      U = rr[1:] * fj
      T = np.linalg.norm(U)
      if N >= mu * T or (T <= 0 and N >= 0):
        pass
      elif mu * N + T <= 0 or (T <= 0 and N < 0):
        cost += 0.5 * np.sum(D[i0:i1] * rr**2); g_r[i0:i1] = D[i0:i1] * rr
      else:
        Dm = D[i0] / (mu * mu * (1 + mu * mu))
        NT = N - mu * T
        cost += 0.5 * Dm * NT**2
        g_r[i0] = Dm * NT * mu
        g_r[i0 + 1:i1] = -Dm * NT * mu * (U / T) * fj
    return cost, M @ (a - a_s) + J.T @ g_r
  return f, M, a_s


def _check_minimiser(m, o, rtol):
  f, M, a_s = _constraint_cost(m, o)
  qacc = np.array(o.qacc)
  c0, g0 = f(qacc)
  # 1. stationarity in the metric of the problem: |grad|_{M^-1} tiny relative to the force scale
  scale = max(1.0, np.abs(M @ a_s).max())
  assert np.sqrt(g0 @ np.linalg.solve(M, g0)) < 1e-5 * scale
  # 2. a generic quasi-Newton minimiser from the unconstrained solution finds no better point, and finds this one
  best = optimize.minimize(lambda a: f(a), a_s, jac=True, method='L-BFGS-B', options=dict(maxiter=5000, ftol=1e-15, gtol=1e-10, maxcor=50))
  assert best.fun >= c0 - 1e-9 * max(1.0, abs(c0)), (best.fun, c0)
  ref = optimize.minimize(lambda a: f(a), qacc, jac=True, method='L-BFGS-B', options=dict(maxiter=5000, ftol=1e-15, gtol=1e-10, maxcor=50))
  np.testing.assert_allclose(ref.x, qacc, rtol=0, atol=rtol * max(1.0, np.abs(qacc).max()))
  return c0


@pytest.mark.parametrize('asset,seed', [('cheetah', 0), ('cheetah', 1), ('humanoid', 0), ('humanoid', 1), ('hopper', 0), ('walker', 0)])
def test_newton_answer_minimises_the_documented_cost_pyramidal(asset, seed):
  m = mc.compile_xml(common.read_model(asset + '.xml'))
  o = OraclePhysics(m)
  rs = np.random.RandomState(seed)
  o.qpos[:] = m.qpos0
  o.qpos[-4:] += rs.uniform(-.3, .3, 4)
  if asset == 'humanoid':
    o.qpos[2] = 0.6
  checked = 0
  for t in range(150):
    o.ctrl[:] = rs.uniform(-1, 1, m.nu)
    o.step()
    if t % 15 == 14 and o.nefc > 0:
      o.forward()
      _check_minimiser(m, o, 2e-6)
      checked += 1
  assert checked >= 5


@pytest.mark.parametrize('asset', ['cheetah', 'hopper'])
def test_cg_answer_minimises_the_documented_cost(asset):
  """option solver="CG": a different iteration on the SAME documented cost -- the generic minimiser must land on
  its answer too (the restated Polak-Ribiere / M^-1 preconditioning is otherwise pinned only against itself)."""
  xml = common.read_model(asset + '.xml').replace('<option', '<option solver="CG" iterations="200" tolerance="1e-10"', 1)
  m = mc.compile_xml(xml)
  assert m.opt.solver == 1
  o = OraclePhysics(m)
  rs = np.random.RandomState(3)
  o.qpos[:] = m.qpos0
  o.qpos[-4:] += rs.uniform(-.3, .3, 4)
  checked = 0
  for t in range(120):
    o.ctrl[:] = rs.uniform(-1, 1, m.nu)
    o.step()
    if t % 15 == 14 and o.nefc > 0:
      o.forward()
      _check_minimiser(m, o, 2e-5)
      checked += 1
  assert checked >= 4


_ELL = """<mujoco><option cone="elliptic" impratio="{imp}" gravity="1.5 .4 -9.81"/>
<default><geom friction=".8 .03 .002" condim="{cd}"/></default><worldbody>
 <geom type="plane" size="3 3 .1"/>
 <body pos="0 0 .15"><freejoint/><geom type="box" size=".1 .07 .12"/>
   <body pos=".1 0 .1"><joint type="hinge" axis="0 1 0" range="-50 50" limited="true" frictionloss=".3"/><geom type="capsule" fromto="0 0 0 .3 0 0" size=".04"/></body></body>
 <body pos=".5 .3 .1"><freejoint/><geom size=".1"/></body>
</worldbody></mujoco>"""


@pytest.mark.parametrize('condim,impratio', [(3, 1.0), (3, 10.0), (4, 2.0), (6, 1.0)])
def test_newton_answer_minimises_the_documented_cost_elliptic_and_friction_loss(condim, impratio):
  """Elliptic three-zone cost (rows 22, 23) and the Huber cost of dof friction loss (row 24), from their definitions."""
  m = mc.compile_xml(_ELL.format(cd=condim, imp=impratio))
  o = OraclePhysics(m)
  rs = np.random.RandomState(condim)
  o.qvel[:] = rs.uniform(-1, 1, m.nv)
  zones = set()
  for t in range(120):
    o.step()
    if t % 10 == 9 and o.nefc:
      o.forward()
      _check_minimiser(m, o, 5e-6)
      zones |= set(np.array(o.efc_state).tolist())
  assert 1 in zones and (2 in zones or condim != 3)      # quadratic rows, and (sliding box) the cone's middle zone


def test_warm_start_choice_and_joint_solve_do_not_change_the_answer():
  """Rows 7 / 8: with any warm start (kept or rejected) the solve ends at the same minimiser, and two mechanically
  independent islands solved jointly give what each gives alone."""
  xml1 = "<body name='a{k}' pos='{x} 0 .2'><freejoint/><geom type='box' size='.1 .1 .1'/><body pos='.1 0 .1'><joint type='hinge' axis='0 1 0'/><geom type='capsule' fromto='0 0 0 .3 0 0' size='.04'/></body></body>"
  both = mc.compile_xml("<mujoco><worldbody><geom type='plane' size='5 5 .1'/>%s%s</worldbody></mujoco>" % (xml1.format(k=0, x=0), xml1.format(k=1, x=2)))
  one = mc.compile_xml("<mujoco><worldbody><geom type='plane' size='5 5 .1'/>%s</worldbody></mujoco>" % xml1.format(k=0, x=0))
  ob, oo = OraclePhysics(both), OraclePhysics(one)
  rs = np.random.RandomState(0)
  v = rs.uniform(-1, 1, one.nv)
  ob.qvel[:one.nv] = v; ob.qvel[one.nv:] = -v[::-1]
  oo.qvel[:] = v
  for t in range(60):
    ob.step(); oo.step()
  np.testing.assert_allclose(np.array(ob.qpos)[:one.nq], np.array(oo.qpos), rtol=0, atol=1e-9)
  ob.forward()
  want = np.array(ob.qacc).copy()
  for trial in range(4):
    ob.qacc_warmstart[:] = want + rs.uniform(-50, 50, both.nv) * (trial > 0) * 10.0**(trial - 2)
    ob.forward()
    np.testing.assert_allclose(np.array(ob.qacc), want, rtol=0, atol=1e-7 * max(1, np.abs(want).max()))


# ---- row 5: invweight0 from finite-difference Jacobians ------------------------------------------------------------
def _fd_invweight(m):
  """body_invweight0 (translational, rotational) and dof_invweight0 at qpos0 from their definition, with body
  Jacobians obtained by finite differences of the kinematics and M = sum_b J_b' diag(m, I_world) J_b."""
  o = OraclePhysics(m)
  nv, nb, eps = m.nv, m.nbody, 1e-6

  def kin(dq):
    o.qpos[:] = m.qpos0
    # integrate a velocity-space displacement dq into qpos (free / ball joints through their quaternions)
    for j in range(m.njnt):
      t, qa, da = int(m.jnt_type[j]), int(m.jnt_qposadr[j]), int(m.jnt_dofadr[j])
      if t in (2, 3):
        o.qpos[qa] += dq[da]
      else:
        if t == 0:
          o.qpos[qa:qa + 3] += dq[da:da + 3]
          qa, da = qa + 3, da + 3
        w = dq[da:da + 3]; ang = np.linalg.norm(w)
        q = np.array(o.qpos[qa:qa + 4])
        if ang > 0:
          ax = w / ang; dqt = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax]
          o.qpos[qa:qa + 4] = np.r_[q[0]*dqt[0] - q[1:] @ dqt[1:], q[0]*dqt[1:] + dqt[0]*q[1:] + np.cross(q[1:], dqt[1:])]
    o.forward()
    return np.array(o.xipos).reshape(nb, 3).copy(), np.array(o.ximat).reshape(nb, 3, 3).copy()
  p0, R0 = kin(np.zeros(nv))
  Jp, Jr = np.zeros((nb, 3, nv)), np.zeros((nb, 3, nv))
  for k in range(nv):
    dq = np.zeros(nv); dq[k] = eps
    p1, R1 = kin(dq)
    Jp[:, :, k] = (p1 - p0) / eps
    for b in range(nb):
      dR = R1[b] @ R0[b].T
      Jr[b, :, k] = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / (2 * eps)
  M = np.zeros((nv, nv))
  for b in range(1, nb):
    Iw = R0[b] @ np.diag(np.asarray(m.body_inertia)[b]) @ R0[b].T
    M += m.body_mass[b] * Jp[b].T @ Jp[b] + Jr[b].T @ Iw @ Jr[b]
  M += np.diag(np.asarray(m.dof_armature))
  Minv = np.linalg.inv(M)
  body = np.array([[np.trace(Jp[b] @ Minv @ Jp[b].T) / 3, np.trace(Jr[b] @ Minv @ Jr[b].T) / 3] for b in range(nb)])
  dof = np.diag(Minv).copy()
  for j in range(m.njnt):
    t, da = int(m.jnt_type[j]), int(m.jnt_dofadr[j])
    if t == 0:
      dof[da:da + 3] = dof[da:da + 3].mean(); dof[da + 3:da + 6] = dof[da + 3:da + 6].mean()
    elif t == 1:
      dof[da:da + 3] = dof[da:da + 3].mean()
  return body, dof, M


@pytest.mark.parametrize('asset', ['cheetah', 'humanoid', 'quadruped', 'cmu_2019_position_floor', 'soccer_2v2_boxhead'])
def test_invweight0_matches_the_definition(asset):
  xml = common.read_model(asset + '.xml') if asset != 'quadruped' else __import__('dm_control_amd.suite.quadruped', fromlist=['x']).make_model()
  m = mc.compile_xml(xml)
  body, dof, M = _fd_invweight(m)
  o = OraclePhysics(m)
  o.qpos[:] = m.qpos0
  o.forward()
  np.testing.assert_allclose(np.array(o.qM).reshape(m.nv, m.nv), M, rtol=0, atol=2e-5 * np.abs(M).max())      # CRB mass matrix
  got_b = np.asarray(m.body_invweight0).reshape(-1, 2)
  sel = np.asarray(m.body_mass) > 0
  sel[0] = False
  np.testing.assert_allclose(got_b[sel], body[sel], rtol=2e-4, atol=1e-9)
  np.testing.assert_allclose(np.asarray(m.dof_invweight0), dof, rtol=2e-4, atol=1e-9)
  assert abs(m.stat_meaninertia - np.diag(M).mean()) < 1e-4 * np.diag(M).mean() if hasattr(m, 'stat_meaninertia') else True


# ---- rows 13 / 33: colliders against brute-force sampling ----------------------------------------------------------
def _two_geoms(g1, g2):
  xml = "<mujoco><option gravity='0 0 0'/><worldbody><body>%s</body><body><freejoint/>%s</body></worldbody></mujoco>" % (g1, g2)
  return mc.compile_xml(xml)


def _contacts(m, q):
  o = OraclePhysics(m)
  o.qpos[:] = q
  o.forward()
  return [o.contact(i) for i in range(o.ncon)], o


def _box_dist(p, half):
  """distance from points p (n, 3) in box coordinates to the solid box (0 inside)."""
  return np.linalg.norm(np.maximum(np.abs(p) - half, 0), axis=1)


def _quat(rs):
  q = rs.randn(4)
  return q / np.linalg.norm(q)


def _rot(q):
  w, x, y, z = q
  return np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])


def test_capsule_capsule_is_the_closest_point_pair_of_the_segments():
  rs = np.random.RandomState(0)
  m = _two_geoms("<geom type='capsule' size='.05 .3' margin='.5'/>", "<geom type='capsule' size='.07 .25' margin='.5'/>")
  t = np.linspace(-1, 1, 801)
  n = 0
  for trial in range(200):
    p, q = rs.uniform(-.5, .5, 3), _quat(rs)
    cons, o = _contacts(m, np.r_[p, q])
    R = _rot(q)
    A = np.outer(t * .3, [0, 0, 1.0]); Bp = p + np.outer(t * .25, R[:, 2])
    d = np.linalg.norm(A[:, None, :] - Bp[None, :, :], axis=2)
    dmin = d.min() - 0.12
    if dmin < 0.45:      # inside the margin with slack for the sampling error
      assert len(cons) >= 1
      assert abs(min(c['dist'] for c in cons) - dmin) < 2e-3, (trial, cons[0]['dist'], dmin)
      i, j = np.unravel_index(d.argmin(), d.shape)
      nrm = (Bp[j] - A[i]) / np.linalg.norm(Bp[j] - A[i])
      c = min(cons, key=lambda c: c['dist'])
      if d.min() > 0.02:
        assert np.dot(c['frame'][0], nrm) > 0.999
      n += 1
  assert n > 100


@pytest.mark.parametrize('other', ['sphere', 'capsule'])
def test_box_pairs_distance_against_surface_sampling(other):
  """Separated configurations: the reported contact distance is the true distance between the two convex sets
  (margin makes separated geoms report).  Box-box is not in this test: like mjc_BoxBox, face contacts report the
  clipped incident polygon against the reference face, which is not a Euclidean distance for far-apart boxes; its
  pin is the penetration-depth test below."""
  rs = np.random.RandomState(1)
  half = np.array([.2, .15, .1])
  g2 = {'sphere': "<geom type='sphere' size='.08' margin='1'/>", 'capsule': "<geom type='capsule' size='.05 .2' margin='1'/>",
        'box': "<geom type='box' size='.12 .1 .08' margin='1'/>"}[other]
  m = _two_geoms("<geom type='box' size='.2 .15 .1' margin='1'/>", g2)
  n = 0
  for trial in range(150):
    p = rs.uniform(-.6, .6, 3); q = _quat(rs); R = _rot(q)
    if other == 'sphere':
      true = _box_dist(p[None], half)[0] - .08
    elif other == 'capsule':
      seg = p + np.outer(np.linspace(-1, 1, 4001) * .2, R[:, 2])
      true = _box_dist(seg, half).min() - .05
    else:
      u = np.linspace(-1, 1, 41)
      faces = []
      for ax in range(3):
        for s in (-1, 1):
          g = np.stack(np.meshgrid(u, u, indexing='ij'), -1).reshape(-1, 2)
          pts = np.insert(g, ax, s, axis=1) * np.array([.12, .1, .08])
          faces.append(pts)
      surf = p + np.concatenate(faces) @ R.T
      true = _box_dist(surf, half).min()
    if true < 0.03:          # penetrating / touching: sampled distances saturate at 0
      continue
    cons, o = _contacts(m, np.r_[p, q])
    assert cons, (trial, true)
    got = min(c['dist'] for c in cons)
    tol = 2e-3 if other != 'box' else 6e-3          # face sampling resolution
    assert abs(got - true) < tol, (other, trial, got, true)
    n += 1
  assert n > 60


def test_box_box_penetration_depth_is_the_minimum_translation_along_the_reported_normal():
  """Overlapping boxes: moving box 2 by (-dist) along the reported normal separates them (to within tolerance),
  and no smaller translation along that normal does: the definition of the penetration depth of the deepest contact."""
  rs = np.random.RandomState(2)
  m = _two_geoms("<geom type='box' size='.2 .15 .1'/>", "<geom type='box' size='.12 .1 .08'/>")
  n = 0
  for trial in range(120):
    p = rs.uniform(-.22, .22, 3); q = _quat(rs)
    cons, _ = _contacts(m, np.r_[p, q])
    if not cons:
      continue
    c = min(cons, key=lambda c: c['dist'])
    if c['dist'] > -5e-3:
      continue
    nrm, depth = c['frame'][0], -c['dist']
    sep, _ = _contacts(m, np.r_[p + nrm * (depth + 2e-3), q])
    assert not sep, (trial, depth)
    still, _ = _contacts(m, np.r_[p + nrm * (depth - 2e-3), q])
    assert still, (trial, depth)
    n += 1
  assert n > 40
