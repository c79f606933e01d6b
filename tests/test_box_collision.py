"""Box pairs: sphere-box, capsule-box, box-box (oracle restatements, PARITY_ASSUMPTIONS.md row 33) pinned on
closed-form configurations and on statics; the kernel core (host build) is compared with the oracle on
random piles."""
import numpy as np
import pytest

from dm_control_amd import mjcf_compiler as mc
from emu_lib import EmuPhysics
from oracle.oracle import OraclePhysics


def _scene(bodies, option=''):
  out = ['<mujoco><option gravity="0 0 -9.81" timestep="0.002" %s/><worldbody><geom name="floor" type="plane" size="3 3 .1"/>' % option]
  for i, (gtype, size, pos, quat) in enumerate(bodies):
    out.append('<body pos="%s" quat="%s"><freejoint/><geom name="g%d" type="%s" size="%s" density="1000"/></body>' % (
        ' '.join(repr(float(x)) for x in pos), ' '.join(repr(float(x)) for x in quat), i, gtype,
        ' '.join(repr(float(x)) for x in size)))
  out.append('</worldbody></mujoco>')
  return '\n'.join(out)


def _contacts(xml):
  o = OraclePhysics(mc.compile_xml(xml))
  o.forward()
  m = o.model.compiled
  return o, [o.contact(i) for i in range(o.ncon) if m.geom_type[o.contact(i)['geom1']] != 0]


I = (1, 0, 0, 0)


def test_sphere_box_closed_forms():
  box = ('box', (.2, .1, .05), (0, 0, 1), I)
  # above the top face
  _, c = _contacts(_scene([('sphere', (.03,), (.05, .02, 1.07), I), box]))
  assert len(c) == 1 and abs(c[0]['dist'] - (.07 - .05 - .03)) < 1e-12
  np.testing.assert_allclose(c[0]['frame'][0], [0, 0, -1], atol=1e-12)       # sphere (geom 1) -> box (geom 2)
  np.testing.assert_allclose(c[0]['pos'], [.05, .02, 1.05 + 0.5*c[0]['dist']], atol=1e-12)
  # off an edge: closest point is on the edge x = .2, z = .05
  _, c = _contacts(_scene([('sphere', (.03,), (.21, 0, 1.06), I), box]))
  d = np.hypot(.01, .01)
  assert abs(c[0]['dist'] - (d - .03)) < 1e-12
  np.testing.assert_allclose(c[0]['frame'][0], [-.01/d, 0, -.01/d], atol=1e-12)
  # off a corner
  _, c = _contacts(_scene([('sphere', (.03,), (.21, .11, 1.06), I), box]))
  assert abs(c[0]['dist'] - (np.sqrt(3)*.01 - .03)) < 1e-12
  # centre inside the box: pushed out through the nearest face (+z here: 1 cm below the top)
  _, c = _contacts(_scene([('sphere', (.03,), (.05, .02, 1.04), I), box]))
  assert abs(c[0]['dist'] - (-.01 - .03)) < 1e-12
  np.testing.assert_allclose(c[0]['frame'][0], [0, 0, -1], atol=1e-12)
  # rotated box: same numbers in the box frame
  q = np.array([.8, .1, .5, .3]); q /= np.linalg.norm(q)
  w, x, y, z = q
  R = np.array([[1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y)], [2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x)],
                [2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y)]])
  centre = np.array([0, 0, 1.]) + R @ [.05, .02, .07]
  _, c = _contacts(_scene([('sphere', (.03,), centre, I), ('box', (.2, .1, .05), (0, 0, 1), q)]))
  assert abs(c[0]['dist'] - (-.01)) < 1e-12
  np.testing.assert_allclose(c[0]['frame'][0], -R[:, 2], atol=1e-12)


def test_capsule_box_contacts():
  box = ('box', (.2, .1, .05), (0, 0, 1), I)
  qy = (np.sqrt(.5), 0, np.sqrt(.5), 0)          # capsule axis along x
  # lying on the top face: both end caps touch
  _, c = _contacts(_scene([('capsule', (.02, .1), (0, 0, 1.065), qy), box]))
  assert len(c) == 2
  np.testing.assert_allclose([k['dist'] for k in c], -.005, atol=1e-12)
  np.testing.assert_allclose(sorted(k['pos'][0] for k in c), [-.1, .1], atol=1e-9)
  # standing on it: one contact under the lower cap
  _, c = _contacts(_scene([('capsule', (.02, .1), (0, 0, 1.165), I), box]))
  assert len(c) == 1 and abs(c[0]['dist'] + .005) < 1e-12
  np.testing.assert_allclose(c[0]['frame'][0], [0, 0, -1], atol=1e-12)
  # crossing over the edge x = .2 at 45 degrees: the closest axis point is interior, single contact on the edge
  q45 = (np.cos(np.pi/8), 0, -np.sin(np.pi/8), 0)    # axis (-1, 0, 1)/sqrt(2): perpendicular to the offset from the edge
  o, c = _contacts(_scene([('capsule', (.02, .3), (.2 + .01, 0, 1.05 + .01), q45), box]))
  assert len(c) == 1
  assert abs(c[0]['dist'] - (np.hypot(.01, .01) - .02)) < 1e-9
  np.testing.assert_allclose(c[0]['pos'][1], 0, atol=1e-9)


def test_box_box_face_edge_and_corner_contacts():
  big = ('box', (.2, .2, .1), (0, 0, 1), I)
  # small box resting 2 mm deep on the big one: its four bottom corners
  _, c = _contacts(_scene([big, ('box', (.05, .04, .03), (.02, .01, 1.128), I)]))
  assert len(c) == 4
  np.testing.assert_allclose([k['dist'] for k in c], -.002, atol=1e-12)
  np.testing.assert_allclose(sorted((round(k['pos'][0], 9), round(k['pos'][1], 9)) for k in c),
                             sorted([(.07, .05), (.07, -.03), (-.03, .05), (-.03, -.03)]), atol=1e-9)
  for k in c:
    np.testing.assert_allclose(k['frame'][0], [0, 0, 1], atol=1e-12)      # box 1 (below) -> box 2 (above)
    assert abs(k['pos'][2] - 1.099) < 1e-12
  # yawed 45 degrees and overhanging a corner of the big box: the clipped polygon has more than 4 vertices, 4 kept
  qz = (np.cos(np.pi/8), 0, 0, np.sin(np.pi/8))
  _, c = _contacts(_scene([big, ('box', (.1, .1, .03), (.15, .15, 1.128), qz)]))
  assert len(c) == 4 and all(abs(k['dist'] + .002) < 1e-12 for k in c)
  pts = np.array([k['pos'][:2] for k in c])
  assert (np.abs(pts) <= .2 + 1e-9).all()                                   # inside the reference face
  assert np.ptp(pts[:, 0]) > .05 and np.ptp(pts[:, 1]) > .05                # spread, not clustered
  # edge against edge: lower box rolled 45 degrees about x (top edge along x), upper box 45 degrees about y
  qx, qy = (np.cos(np.pi/8), np.sin(np.pi/8), 0, 0), (np.cos(np.pi/8), 0, np.sin(np.pi/8), 0)
  h = .1*np.sqrt(2)
  _, c = _contacts(_scene([('box', (.3, .1, .1), (0, 0, 1), qx), ('box', (.1, .3, .1), (0, 0, 1 + 2*h - .004), qy)]))
  assert len(c) == 1 and abs(c[0]['dist'] + .004) < 1e-9
  np.testing.assert_allclose(c[0]['frame'][0], [0, 0, 1], atol=1e-9)
  np.testing.assert_allclose(c[0]['pos'], [0, 0, 1 + h - .002], atol=1e-9)
  # a corner poked into a face
  qc = np.array([np.cos(.4775), np.sin(.4775)*np.sqrt(.5), -np.sin(.4775)*np.sqrt(.5), 0])   # (1,1,1) diagonal down
  o, c = _contacts(_scene([big, ('box', (.05, .05, .05), (0, 0, 1.1 + .05*np.sqrt(3) - .003), qc)]))
  assert len(c) >= 1 and min(k['dist'] for k in c) < -.002
  # separated by more than the margin: nothing
  _, c = _contacts(_scene([big, ('box', (.05, .04, .03), (.02, .01, 1.1301), I)]))
  assert len(c) == 0


def test_tower_of_boxes_stands_and_weighs_right():
  bodies = [('box', (.12, .12, .04), (0, 0, .04), I), ('box', (.1, .08, .04), (.01, 0, .12), (np.cos(.2), 0, 0, np.sin(.2))),
            ('box', (.06, .06, .04), (0, .01, .20), I), ('sphere', (.03,), (0, 0, .27), I)]
  m = mc.compile_xml(_scene(bodies, 'cone="elliptic"').replace('density="1000"/></body>', 'density="1000" condim="6" friction="1 .01 .01"/></body>'))
  o, e = OraclePhysics(m), EmuPhysics(m, 64, nconmax=32)
  o.forward()
  for _ in range(2000):
    o.step()
    e.step()
  assert np.abs(o.qvel).max() < 5e-3          # the ball on top is still creeping to its rest point
  np.testing.assert_allclose(o.qpos[[2, 9, 16, 23]], [.04, .12, .20, .27], atol=2e-3)      # still a tower
  floor = sum(o.contact_force(i)[0, 0] for i in range(o.ncon) if o.contact(i)['geom1'] == 0)
  np.testing.assert_allclose(floor, m.body_mass[1:].sum() * 9.81, rtol=1e-4)   # not perfectly static yet
  np.testing.assert_allclose(e.qpos, o.qpos, atol=1e-9)
  assert not o.warning.any() and not e.warning.any()


@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('prec,tol', [(64, 1e-9), (32, 5e-4)])
def test_random_pile_kernel_core_matches_oracle(seed, prec, tol):
  rs = np.random.RandomState(seed)
  bodies = []
  for k in range(5):
    gtype = ['box', 'box', 'capsule', 'sphere', 'box'][k]
    size = {'box': rs.uniform(.04, .12, 3), 'capsule': (rs.uniform(.03, .05), rs.uniform(.05, .1)), 'sphere': (rs.uniform(.04, .07),)}[gtype]
    q = rs.randn(4)
    bodies.append((gtype, size, (rs.uniform(-.08, .08), rs.uniform(-.08, .08), .15 + .22*k), q/np.linalg.norm(q)))
  m = mc.compile_xml(_scene(bodies, 'cone="%s"' % ('elliptic' if seed % 2 else 'pyramidal')))
  o, e = OraclePhysics(m), EmuPhysics(m, prec, nconmax=32)
  o.forward()
  pairs = set()
  for _ in range(300):               # the pile forms; later it is as chaotic as any pile
    o.step()
    e.step()
    for i in range(o.ncon):
      c = o.contact(i)
      pairs.add((int(m.geom_type[c['geom1']]), int(m.geom_type[c['geom2']])))
  assert any(p[1] == 6 and p[0] != 0 for p in pairs)      # some box pair was exercised
  np.testing.assert_allclose(e.qpos, o.qpos, atol=tol)
  assert not o.warning.any() and not e.warning.any()
