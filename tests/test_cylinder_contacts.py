"""A sphere or a capsule against a CYLINDER (MuJoCo: mjc_SphereCylinder is analytic, capsule-cylinder goes through its
convex collider and yields the closest pair; here both are exact geometry on the solid cylinder -- PARITY_ASSUMPTIONS row
42).  The other cylinder pairs (cylinder-cylinder, box / ellipsoid against a cylinder) stay guarded: a hit of the
enclosing capsule raises dmcWARN_COLLISION instead of producing a contact."""
import numpy as np
import pytest

from dm_control_amd import mjcf_compiler as mc
from oracle.oracle import OracleModel, OraclePhysics

_XML = """<mujoco><option timestep='0.002'/><worldbody>
<geom name='floor' type='plane' size='3 3 .1'/>
<geom name='cyl' type='cylinder' size='.5 .2' pos='0 0 .2' %s/>
<body name='b' pos='%s' %s><freejoint/><geom name='g' type='%s' size='%s'/></body>
</worldbody></mujoco>"""


def _oracle(xml):
  p = OraclePhysics(OracleModel(mc.compile_xml(xml)))
  p.forward()
  return p


def _cyl_contacts(p):
  out = []
  for i in range(p.ncon):
    c = p.contact(i)
    if 1 in (int(c['geom1']), int(c['geom2'])):      # geom 1 is the cylinder
      out.append(c)
  return out


@pytest.mark.parametrize('pos,dist,normal,point', [
    ('0 0 .49', -0.01, (0, 0, -1), (0, 0, 0.395)),                        # on the cap
    ('.59 0 .2', -0.01, (-1, 0, 0), (0.495, 0, 0.2)),                     # against the side
    ('.55 0 .45', 0.05 * np.sqrt(2) - 0.1, (-np.sqrt(.5), 0, -np.sqrt(.5)), None),      # on the rim
    ('.1 0 .25', -0.25, (0, 0, -1), None),                                # centre inside: out through the nearer face (the cap)
    ('.45 0 .2', -0.15, (-1, 0, 0), None),                                # centre inside, nearer to the side: the normal (sphere -> cylinder) points at the axis
])
def test_sphere_against_cylinder_cases(pos, dist, normal, point):
  p = _oracle(_XML % ('', pos, '', 'sphere', '.1'))
  cs = _cyl_contacts(p)
  assert len(cs) == 1 and not p.warning.any()
  c = cs[0]
  assert int(c['geom1']) == 2 and int(c['geom2']) == 1      # sphere (type 2) sorts before the cylinder (type 5)
  np.testing.assert_allclose(c['dist'], dist, atol=1e-12)
  np.testing.assert_allclose(c['frame'][0], normal, atol=1e-12)
  if point is not None:
    np.testing.assert_allclose(c['pos'], point, atol=1e-12)


@pytest.mark.parametrize('pos,attitude,dist,normal,point', [
    ('0 0 .49', "euler='0 90 0'", -0.01, (0, 0, -1), (0, 0, 0.395)),            # lying on the cap: the middle of the flat stretch
    ('.59 0 .2', '', -0.01, (-1, 0, 0), (0.495, 0, 0.2)),                       # alongside the cylinder
    ('.75 0 .49', "euler='0 90 0'", -0.01, (0, 0, -1), (0.475, 0, 0.395)),      # one end over the cap: middle of the part above it
    ('.9 0 .7', "euler='0 45 0'", None, None, None),                            # tilted above the rim: closest pair on the rim
])
def test_capsule_against_cylinder_cases(pos, attitude, dist, normal, point):
  p = _oracle(_XML % ('', pos, attitude, 'capsule', '.1 .3'))
  cs = _cyl_contacts(p)
  if dist is None:
    # compare with a brute-force search over the axis segment
    R = np.array(p.geom_xmat).reshape(-1, 3, 3)[2]
    c0 = np.array(p.geom_xpos).reshape(-1, 3)[2]
    best = _closest_over_axis(c0, R[:, 2], .3, 2000001)
    if best - 0.1 > 0:
      assert not cs
    else:
      np.testing.assert_allclose(cs[0]['dist'], best - 0.1, atol=1e-9)
    return
  assert len(cs) == 1 and not p.warning.any()
  np.testing.assert_allclose(cs[0]['dist'], dist, atol=1e-9)
  np.testing.assert_allclose(cs[0]['frame'][0], normal, atol=1e-7)
  np.testing.assert_allclose(cs[0]['pos'], point, atol=1e-6)


def _point_cyl(q, p=np.array([0, 0, .2]), a=np.array([0, 0, 1.0]), R=.5, H=.2):
  """distance of the points q (n, 3) to the solid cylinder"""
  v = np.atleast_2d(q) - p
  x = v @ a
  perp = v - x[:, None] * a
  d = np.linalg.norm(perp, axis=1)
  scale = np.where(d > R, R / np.maximum(d, 1e-300), 1.0)
  closest = p + np.clip(x, -H, H)[:, None] * a + perp * scale[:, None]
  return np.linalg.norm(np.atleast_2d(q) - closest, axis=1)


def _closest_over_axis(c0, axis, h, n):
  t = np.linspace(-h, h, n)
  return float(_point_cyl(c0 + t[:, None] * axis).min())


def test_capsule_axis_through_the_cylinder_only_warns():
  p = _oracle(_XML % ('', '0 0 .6', '', 'capsule', '.1 .3'))
  assert not _cyl_contacts(p) and p.warning[8] == 1      # dmcWARN_COLLISION


def test_random_capsules_agree_with_a_brute_force_closest_pair():
  rs = np.random.RandomState(0)
  for trial in range(40):
    pos = rs.uniform(-.9, .9, 3) + [0, 0, .6]
    quat = rs.randn(4)
    quat /= np.linalg.norm(quat)
    xml = _XML % ('', '%g %g %g' % tuple(pos), "quat='%g %g %g %g'" % tuple(quat), 'capsule', '.08 .25')
    p = _oracle(xml)
    R = np.array(p.geom_xmat).reshape(-1, 3, 3)[2]
    c0 = np.array(p.geom_xpos).reshape(-1, 3)[2]
    best = _closest_over_axis(c0, R[:, 2], .25, 200001)
    cs = _cyl_contacts(p)
    if best < 1e-6:
      assert p.warning[8] >= 1 or cs == []
    elif best - 0.08 > 1e-6:
      assert not cs
    elif best - 0.08 < -1e-6:
      assert len(cs) == 1
      np.testing.assert_allclose(cs[0]['dist'], best - 0.08, atol=1e-9)      # (the grid search resolves 2.5e-6 along the axis)
      # the frame's normal is the unit vector between the closest pair, the point half way between the surfaces
      n = cs[0]['frame'][0]
      np.testing.assert_allclose(np.linalg.norm(n), 1, atol=1e-12)
      q = cs[0]['pos'] - n * (0.08 + 0.5 * cs[0]['dist'])      # back to the axis point
      assert abs(np.linalg.norm(np.cross(q - c0, R[:, 2]))) < 1e-9      # ... which lies on the capsule's axis


def test_sphere_and_capsule_come_to_rest_on_a_cylinder():
  for kind, size, z in (('sphere', '.1', .1), ('capsule', '.1 .2', .1)):
    p = _oracle(_XML % ('', '0.1 0.05 .6', "euler='0 90 0'" if kind == 'capsule' else '', kind, size))
    p.step(1500)
    assert not p.warning.any()
    assert abs(np.array(p.qpos)[2] - (0.4 + z)) < 2e-3, np.array(p.qpos)[:3]      # rests on the cap at z = 0.4
    assert np.abs(np.array(p.qvel)).max() < 1e-3


# ---- kernel core (host emulation) and device against the oracle -------------------------------------------------------
_SCENE = """<mujoco><option timestep='0.002'/><worldbody>
<geom name='floor' type='plane' size='3 3 .1'/>
<geom name='post' type='cylinder' size='.3 .25' pos='0 0 .25'/>
<body name='drum' pos='1.2 0 .5'><joint type='hinge' axis='0 1 0'/><geom name='drumg' type='cylinder' size='.3 .1' euler='90 0 0'/></body>
<body name='s' pos='.05 .02 .7'><freejoint/><geom type='sphere' size='.12'/></body>
<body name='c' pos='-.1 .05 1.1' euler='10 80 0'><freejoint/><geom type='capsule' size='.08 .25'/></body>
<body name='c2' pos='1.15 0 1.0' euler='90 0 20'><freejoint/><geom type='capsule' size='.06 .2'/></body>
<body name='s2' pos='1.3 .02 1.4'><freejoint/><geom type='sphere' size='.1'/></body>
</worldbody></mujoco>"""


# fp32: free bodies tumbling for 600 steps, open loop.  The error stays below 1e-5 for 500 steps; at step 502 a contact is
# just touching and fp32 / fp64 activate it a step apart (MuJoCo's dynamics are discontinuous there), after which the two
# trajectories are 2e-3 apart: the tolerance is the device test's (below), not a statement about the step.
@pytest.mark.parametrize('prec,tol', [(64, 1e-9), (32, 5e-3)])
def test_kernel_core_matches_oracle_with_cylinder_contacts(prec, tol):
  from emu_lib import EmuPhysics

  class Dev:
    def __init__(self, m):
      self.e = EmuPhysics(m, prec=prec, nconmax=32)
      self.e.forward()
    def step(self, n): self.e.step(n, legacy=False)
    def qpos(self): return np.array(self.e.qpos).ravel()
    def ncon(self): return int(np.array(self.e.ncon).ravel()[0])
    def warning(self): return np.array(self.e.warning).ravel()
  # non-legacy oracle stepping to match
  m = mc.compile_xml(_SCENE)
  o = OraclePhysics(OracleModel(m), legacy_step=False)
  o.forward()
  d = Dev(m)
  seen = 0
  for k in range(30):
    o.step(20)
    d.step(20)
    assert not np.array(o.warning).any() and not d.warning().any()
    np.testing.assert_allclose(d.qpos(), np.array(o.qpos), rtol=0, atol=tol)
    if prec == 64:
      assert d.ncon() == int(o.ncon)
    seen = max(seen, sum(1 for i in range(o.ncon) if {int(o.contact(i)['geom1']), int(o.contact(i)['geom2'])} & {1, 2}))
  assert seen >= 3      # the spheres and capsules did land on the two cylinders


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol', [(64, 1e-9), (32, 5e-3)])      # (fp32: free bodies tumbling for 600 steps)
def test_device_matches_oracle_with_cylinder_contacts(precision, tol):
  from dm_control_amd.batch import BatchedPhysics
  m = mc.compile_xml(_SCENE)
  o = OraclePhysics(OracleModel(m), legacy_step=False)
  o.forward()
  b = BatchedPhysics(m, 2, precision=precision, nconmax=32)
  b.legacy_step = False
  b.forward(False)
  for k in range(30):
    o.step(20)
    b.step(20)
    assert not np.array(o.warning).any() and not b.get('warning').any()
    np.testing.assert_allclose(b.get('qpos')[1], np.array(o.qpos), rtol=0, atol=tol)
    if precision == 64:
      assert int(b.get('ncon')[0, 0]) == int(o.ncon)
