"""Constraint islands (mj_island + the per-island solves of mj_fwdConstraint; `<flag island>` is a DISABLE flag that
defaults to enable, mjcf/schema.xml:102).

The kinematic trees joined by constraints form islands; each island is its own convex problem (the cross blocks of M
and of J' D J are exact zeros), solved with its own line search / iteration count / stopping test; trees without a
constraint take qacc_smooth.  MuJoCo solves per island exactly when the flag is on, at least one island exists, no
noslip pass is configured and the solver is CG or Newton (PARITY_ASSUMPTIONS rows 37-39: restated from the published
algorithm, unpinned against a MuJoCo binary).  Pins here: the partition itself; every island's answer against the joint
solve of the same step (the minimiser is unique: 1e-9) and against the tree simulated alone; `island="disable"` and
noslip models keep the one-system solve bit for bit."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dm_control_amd import mjcf_compiler as mc  # noqa: E402

LEG = """
    <body name='{n}' pos='{x} 0 .45'>
      <joint name='{n}_slide' type='slide' axis='0 0 1' damping='.2'/>
      <joint name='{n}_hip' type='hinge' axis='0 1 0' damping='.05' range='-1 1' limited='true'/>
      <geom name='{n}_thigh' type='capsule' fromto='0 0 0 .25 0 -.2' size='.04' mass='1'/>
      <body pos='.25 0 -.2'>
        <joint name='{n}_knee' type='hinge' axis='0 1 0' damping='.05'/>
        <geom name='{n}_shin' type='capsule' fromto='0 0 0 -.1 0 -.25' size='.03' mass='.5' condim='{condim}'/>
      </body>
    </body>
"""


def scene(legs, flags='', option='', condim=3, extra=''):
  bodies = ''.join(LEG.format(n=n, x=x, condim=condim) for n, x in legs)
  acts = ''.join("<motor joint='%s_hip' gear='20'/><motor joint='%s_knee' gear='10'/>" % (n, n) for n, _ in legs)
  return """<mujoco><option timestep='0.004' %s><flag %s/></option><worldbody>
    <geom name='floor' type='plane' size='5 5 .1'/>%s%s</worldbody><actuator>%s</actuator></mujoco>""" % (option, flags, bodies, extra, acts)


def _oracle(xml, q=None, v=None, ctrl=None):
  from oracle.oracle import OraclePhysics
  m = mc.compile_xml(xml)
  p = OraclePhysics(m)
  if q is not None:
    p.qpos[:] = q
  if v is not None:
    p.qvel[:] = v
  if ctrl is not None:
    p.ctrl[:] = ctrl
  return m, p


def _settled_state(n):
  rs = np.random.RandomState(n)
  q = np.tile([-.18, .4, -.9], n) + .2 * rs.randn(3 * n)
  q[0::3] = -.18 + .02 * rs.rand(n)      # low enough for the shin to be in the floor
  return q, rs.randn(3 * n), rs.uniform(-1, 1, 2 * n)


def test_flag_is_a_disable_flag_and_defaults_to_enable():
  assert mc.C['DMC_DSBL_ISLAND'] == 1 << 18
  assert mc.compile_xml(scene([('a', 0)])).opt.disableflags & mc.C['DMC_DSBL_ISLAND'] == 0
  assert mc.compile_xml(scene([('a', 0)], flags="island='disable'")).opt.disableflags & mc.C['DMC_DSBL_ISLAND']


def test_partition_follows_the_trees_that_constraints_join():
  legs = [('a', -1.5), ('b', 0.0), ('c', 1.5)]
  q, v, c = _settled_state(3)
  m, p = _oracle(scene(legs), q, v, c)
  p.forward()
  assert p.ncon >= 3 and p.nisland == 3
  isl = p.dof_island()
  assert list(isl) == [0, 0, 0, 1, 1, 1, 2, 2, 2]
  rows = p.efc_island()
  assert set(rows) == {0, 1, 2} and len(rows) == p.nefc
  # lift the middle leg out of the floor and inside its joint range: its tree has no constraint -> no island, qacc_smooth
  q2 = q.copy(); q2[3] = 0.5; q2[4] = 0.0
  m, p = _oracle(scene(legs), q2, v, c)
  p.forward()
  assert p.nisland == 2 and list(p.dof_island()) == [0, 0, 0, -1, -1, -1, 1, 1, 1]
  np.testing.assert_array_equal(p.qacc[3:6], p.qacc_smooth[3:6])
  # a weld between the outer legs joins their trees into one island; the middle one stays alone
  weld = "<equality><connect body1='a' body2='c' anchor='0 0 0'/></equality>"
  m, p = _oracle(scene(legs).replace('<actuator>', weld + '<actuator>'), q, v, c)
  p.forward()
  assert p.nisland == 2 and list(p.dof_island()) == [0, 0, 0, 1, 1, 1, 0, 0, 0]


@pytest.mark.parametrize('option,condim', [('', 3), ("cone='elliptic'", 3), ("cone='elliptic' impratio='3'", 4), ("solver='CG'", 3)])
def test_island_solves_equal_the_joint_solve_and_the_trees_alone(option, condim):
  legs = [('a', -1.5), ('b', 0.0), ('c', 1.5)]
  q, v, c = _settled_state(3)
  tight = option + " tolerance='1e-12'"
  _, pi = _oracle(scene(legs, option=tight, condim=condim), q, v, c)
  _, pj = _oracle(scene(legs, flags="island='disable'", option=tight, condim=condim), q, v, c)
  pi.forward(); pj.forward()
  assert pi.nisland == 3 and pj.nisland == 0 and pi.nefc == pj.nefc > 6
  np.testing.assert_allclose(pi.qacc, pj.qacc, rtol=0, atol=1e-9 * max(1, np.abs(pj.qacc).max()))
  np.testing.assert_allclose(pi.efc_force[:pi.nefc], pj.efc_force[:pj.nefc], rtol=0, atol=1e-8 * max(1, np.abs(pj.efc_force).max()))
  # each tree simulated alone (one island = the whole system: the in-place solve) gives its island's answer
  for k, (n, x) in enumerate(legs):
    _, pk = _oracle(scene([(n, x)], option=tight, condim=condim), q[3*k:3*k + 3], v[3*k:3*k + 3], c[2*k:2*k + 2])
    pk.forward()
    assert pk.nisland == 1
    np.testing.assert_allclose(pk.qacc, pi.qacc[3*k:3*k + 3], rtol=0, atol=1e-9 * max(1, np.abs(pk.qacc).max()))


def test_noslip_models_and_disabled_flag_keep_the_one_system_solve_bit_for_bit():
  legs = [('a', -1.5), ('b', 0.0)]
  q, v, c = _settled_state(2)
  _, pn = _oracle(scene(legs, option="noslip_iterations='3'"), q, v, c)
  _, pd = _oracle(scene(legs, option="noslip_iterations='3'", flags="island='disable'"), q, v, c)
  pn.forward(); pd.forward()
  assert pn.nisland == 0 and pd.nisland == 0
  np.testing.assert_array_equal(pn.qacc, pd.qacc)
  # a one-tree model: one island that IS the system -- identical to the disabled flag
  q1, v1, c1 = _settled_state(1)
  _, a = _oracle(scene([('a', 0)]), q1, v1, c1)
  _, b = _oracle(scene([('a', 0)], flags="island='disable'"), q1, v1, c1)
  for _ in range(50):
    a.step(); b.step()
  assert a.nisland == 1 and b.nisland == 0
  np.testing.assert_array_equal(a.qpos, b.qpos)


@pytest.mark.parametrize('domain,task', [('manipulator', 'bring_ball'), ('stacker', 'stack_2'), ('finger', 'spin'), ('quadruped', 'fetch')])
def test_suite_multi_tree_models_track_the_one_system_solve(oracle_backend, domain, task):
  """Rollouts of the multi-tree suite models with islands (the default) against the same model with the flag disabled:
  the two solvers reach the same minimiser every step (tolerance 1e-8 scaled), so the trajectories stay together."""
  from dm_control_amd import suite
  from oracle.oracle import OraclePhysics
  env = suite.load(domain, task, task_kwargs=dict(random=3))
  env.reset()
  m = env.physics.model
  mi = m
  import copy
  md = copy.copy(m); md.opt = copy.copy(m.opt); md.opt.disableflags = int(m.opt.disableflags) | mc.C['DMC_DSBL_ISLAND']
  a, b = OraclePhysics(mi), OraclePhysics(md)
  for p in (a, b):
    p.qpos[:] = env.physics.data.qpos; p.qvel[:] = env.physics.data.qvel
    p.forward()
  rs = np.random.RandomState(0)
  seen = 0
  for t in range(500):
    c = rs.uniform(-1, 1, m.nu)
    a.ctrl[:] = c; b.ctrl[:] = c
    a.step(); b.step()
    seen = max(seen, a.nisland)
    assert b.nisland == 0
  err = np.abs(a.qpos - b.qpos).max()
  print('measured: %s %s islands up to %d, |dqpos| after 500 steps %.2e' % (domain, task, seen, err))
  assert err < 1e-6, err
  if domain != 'finger':      # (the finger's spinner is only ever reached through the finger: one island at most)
    assert seen >= 2, seen


# ---- the kernel core and the device -------------------------------------------------------------------------------
@pytest.mark.parametrize('option,condim', [('', 3), ("cone='elliptic' impratio='3'", 4), ("solver='CG'", 3)])
def test_kernel_core_island_solves_follow_the_oracle(option, condim):
  """StepCore::solve_islands (fp64: on by default): the partition and the per-island iterates of the kernel core against
  the oracle's gathered sub-problems over a rollout in which legs touch down and lift off -- qacc to 1e-11 every step
  (the joint solve of the same core, `set_islands(0)`, differs from both at the solver's tolerance, not at rounding)."""
  from emu_lib import EmuPhysics
  from oracle.oracle import OraclePhysics
  legs = [('a', -1.5), ('b', 0.0), ('c', 1.5)]
  xml = scene(legs, option=option, condim=condim)
  m = mc.compile_xml(xml)
  q, v, _ = _settled_state(3)
  e, j, o = EmuPhysics(m, 64), EmuPhysics(m, 64), OraclePhysics(m)
  j.set_islands(0)
  for p in (e, j, o):
    p.qpos[:] = q; p.qvel[:] = v
  o.forward()
  rs = np.random.RandomState(4)
  worst_i = worst_j = 0.0
  most = 0
  for t in range(150):
    c = rs.uniform(-1, 1, m.nu)
    for p in (e, j, o):
      p.ctrl[:] = c
      p.step()
    most = max(most, o.nisland)
    worst_i = max(worst_i, np.abs(e.qpos - o.qpos).max())
    worst_j = max(worst_j, np.abs(j.qpos - o.qpos).max())
  print('measured: kernel core vs island oracle over 150 steps: per-island solves %.2e, joint solve %.2e (islands up to %d)' % (worst_i, worst_j, most))
  assert most == 3
  assert worst_i < 1e-11, worst_i
  assert worst_j < (1e-4 if 'CG' in option else 1e-6), worst_j      # (CG stops at a looser point: its joint and per-island answers differ more)


def test_kernel_core_islands_off_in_fp32_by_default_and_identical_for_one_tree():
  from emu_lib import EmuPhysics
  legs = [('a', -1.5), ('b', 0.0)]
  m = mc.compile_xml(scene(legs))
  q, v, c = _settled_state(2)
  a, b = EmuPhysics(m, 32), EmuPhysics(m, 32)
  b.set_islands(0)
  for p in (a, b):
    p.qpos[:] = q; p.qvel[:] = v; p.ctrl[:] = c
    for _ in range(30):
      p.step()
  np.testing.assert_array_equal(a.qpos, b.qpos)      # fp32: the throughput instantiation solves jointly unless asked
  # one tree: the island that holds everything is the joint problem, solved in place
  m1 = mc.compile_xml(scene([('a', 0)]))
  q1, v1, c1 = _settled_state(1)
  a, b = EmuPhysics(m1, 64), EmuPhysics(m1, 64)
  b.set_islands(0)
  for p in (a, b):
    p.qpos[:] = q1; p.qvel[:] = v1; p.ctrl[:] = c1
    for _ in range(30):
      p.step()
  np.testing.assert_array_equal(a.qpos, b.qpos)


def test_kernel_core_fp32_cg_solves_per_island_by_default():
  """Round 6: the default (`islands` = -1) means per-island solves wherever the joint answer differs
  measurably from MuJoCo's -- fp64, and CG at ANY precision (CG stops at a looser point: 4.9e-6 between the two forms) --
  and the joint solve only for fp32 Newton (8e-15 apart)."""
  from emu_lib import EmuPhysics
  legs = [('a', -1.5), ('b', 0.0), ('c', 1.5)]
  m = mc.compile_xml(scene(legs, option="solver='CG'"))
  q, v, c = _settled_state(3)
  out = {}
  for name, isl in (('default', None), ('on', 1), ('off', 0)):
    p = EmuPhysics(m, 32)
    if isl is not None: p.set_islands(isl)
    p.qpos[:] = q; p.qvel[:] = v; p.ctrl[:] = c
    for _ in range(30):
      p.step()
    out[name] = p.qpos.copy()
  np.testing.assert_array_equal(out['default'], out['on'])
  assert np.abs(out['default'] - out['off']).max() > 0      # (the joint CG solve is a different iterate sequence)


@pytest.mark.gpu
def test_device_island_solves_follow_the_oracle():
  """The same on the device, fp64 (islands on by default) and fp32 with the option switched on, against per-environment
  oracles; `islands = 0` gives the joint solve."""
  from dm_control_amd.batch import BatchedPhysics
  from oracle.oracle import OraclePhysics, OracleModel
  legs = [('a', -1.5), ('b', 0.0), ('c', 1.5)]
  m = mc.compile_xml(scene(legs, option="cone='elliptic'"))
  B = 6
  rs = np.random.RandomState(8)
  q = np.stack([_settled_state(3)[0] + 0.05 * rs.randn(9) for _ in range(B)])
  om = OracleModel(m)
  refs = [OraclePhysics(om) for _ in range(B)]
  for k, o in enumerate(refs):
    o.qpos[:] = q[k]; o.forward()
  b64, b32, bj = BatchedPhysics(m, B, precision=64), BatchedPhysics(m, B, precision=32), BatchedPhysics(m, B, precision=64)
  b32.set_opt('islands', 1); bj.set_opt('islands', 0)
  for b in (b64, b32, bj):
    b.set('qpos', q)
  most = 0
  for t in range(120):
    c = rs.uniform(-1, 1, (B, m.nu))
    for b in (b64, b32, bj):
      b.set('ctrl', c); b.step()
    for k, o in enumerate(refs):
      o.ctrl[:] = c[k]; o.step(); most = max(most, o.nisland)
  qo = np.stack([o.qpos for o in refs])
  assert most == 3
  np.testing.assert_allclose(b64.get('qpos'), qo, rtol=0, atol=1e-10)
  np.testing.assert_allclose(bj.get('qpos'), qo, rtol=0, atol=1e-6)
  np.testing.assert_allclose(b32.get('qpos'), qo, rtol=0, atol=2e-3)
  for b in (b64, b32, bj):
    assert not b.get('warning').any()
    b.close()


ORTHOGONAL = """<mujoco><option timestep='0.004' cone='elliptic' impratio='2'/><worldbody>
  <body name='cart' pos='0 0 .1'>
    <joint name='cart_x' type='slide' axis='1 0 0' damping='.1'/>
    <geom name='deck' type='box' size='.5 .3 .1' mass='2' friction='.6 .005 .0001'/>
  </body>
  <body name='ball' pos='0 0 .29'>
    <joint name='ball_z' type='slide' axis='0 0 1'/>
    <geom name='ball' type='sphere' size='.1' mass='1' friction='.6 .005 .0001'/>
  </body>
</worldbody><actuator><motor joint='cart_x' gear='10'/></actuator></mujoco>"""


def test_one_contact_whose_rows_move_disjoint_trees_is_one_island():
  """An elliptic contact between a cart that can only slide along x and a ball that can only move along z: the NORMAL row
  moves only the ball's tree, the TANGENT row only the cart's.  mj_island unites the trees over the whole row group of a
  contact, so this is ONE island (the friction on the cart is bounded by the normal force on the ball).  The kernel joins
  the trees AFTER the per-contact union of the row masks; joining them from each row's own dofs left two
  components that both owned every row of the contact -- measured here before the fix: the same answer to 6e-17, because a
  parked dof is a start value and not a constraint, so each "island" solve was the joint problem over again (twice the
  work, not a wrong cone)."""
  from emu_lib import EmuPhysics
  from oracle.oracle import OraclePhysics
  m = mc.compile_xml(ORTHOGONAL)
  e, j, o = EmuPhysics(m, 64), EmuPhysics(m, 64), OraclePhysics(m)
  j.set_islands(0)
  o.forward()
  worst_e = worst_j = 0.0
  most = 0
  for t in range(200):
    c = [np.sin(0.05 * t)]
    for p in (e, j, o):
      p.ctrl[:] = c
      p.step()
    most = max(most, o.nisland)
    worst_e = max(worst_e, np.abs(e.qpos - o.qpos).max())
    worst_j = max(worst_j, np.abs(j.qpos - o.qpos).max())
  print('measured: orthogonal sliders, kernel core vs oracle %.2e (islands on), %.2e (joint solve); islands %d' % (worst_e, worst_j, most))
  assert most == 1
  assert abs(o.qpos[0]) > 1e-3      # the cart did move: the tangent row carried force
  assert worst_e < 1e-10, worst_e
  assert worst_j < 1e-10, worst_j
