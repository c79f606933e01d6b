"""Host-logic test: the kernel core (dm_control_amd/csrc/step_core.h) compiled for
the CPU with one lane per environment (tests/emu) against the fp64 oracle.  This
checks indexing / algorithm logic without a GPU; the real parity tests are the
`-m gpu` ones that call the HIP kernel through the C-ABI."""
import os

import numpy as np
import pytest

from dm_control_amd import mjcf_compiler as mc
from emu_lib import EmuPhysics
from oracle.oracle import OraclePhysics

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      'dm_control_amd', 'suite', 'assets')


def _cheetah():
  with open(os.path.join(ASSETS, 'cheetah.xml')) as f:
    return mc.compile_xml(f.read())


def test_forward_stages_fp64():
  m = _cheetah()
  rs = np.random.RandomState(0)
  for trial in range(4):
    o, e = OraclePhysics(m), EmuPhysics(m, 64)
    q = m.qpos0.copy()
    q[2:] += rs.uniform(-0.3, 0.3, 7)
    q[1] = rs.uniform(-0.6, 0.0)
    v, c = rs.uniform(-1, 1, 9), rs.uniform(-1, 1, 6)
    o.qpos[:], o.qvel[:], o.ctrl[:] = q, v, c
    e.qpos[:], e.qvel[:], e.ctrl[:] = q, v, c
    o.forward()
    e.forward()
    assert o.ncon == e.ncon[0] and o.nefc == e.nefc[0]
    # kinematics is re-associated in the kernel (parallel local poses + composition),
    # so downstream stages agree to rounding, not bit for bit
    for name in ('xpos', 'xmat', 'subtree_com', 'qfrc_bias', 'sensordata'):
      np.testing.assert_allclose(getattr(o, name), getattr(e, name), rtol=1e-12, atol=1e-12, err_msg=name)
    ne = o.nefc
    np.testing.assert_allclose(o.qM, e.dense_M(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(o.efc_J[:ne*m.nv], e.dense_J(), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(o.efc_aref[:ne], e.scratch('efc_aref')[:ne], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(o.qacc, e.qacc, rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize('prec,tol', [(64, 1e-10), (32, 1e-4)])
def test_trajectory_1000_steps(prec, tol):
  # north_star tolerance: 1e-4 rel qpos error over 1000 steps
  m = _cheetah()
  o, e = OraclePhysics(m), EmuPhysics(m, prec)
  rs = np.random.RandomState(1)
  q = m.qpos0.copy()
  q[3:] += rs.uniform(-0.3, 0.3, 6)
  o.qpos[:] = q
  e.qpos[:] = q
  o.forward()
  worst = 0.0
  for _ in range(1000):
    c = rs.uniform(-1, 1, 6)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    worst = max(worst, np.abs(o.qpos - e.qpos).max() / max(1, np.abs(o.qpos).max()))
  assert worst < tol
  assert not e.warning.any()


def test_nstep_fused_equals_single_steps():
  m = _cheetah()
  a, b = EmuPhysics(m, 64), EmuPhysics(m, 64)
  a.ctrl[:] = 0.3
  b.ctrl[:] = 0.3
  for _ in range(5):
    a.step(1)
  b.step(5)
  np.testing.assert_array_equal(a.qpos, b.qpos)
  np.testing.assert_array_equal(a.sensordata, b.sensordata)


def test_contact_cap_raises_warning():
  m = _cheetah()
  e = EmuPhysics(m, 64, nconmax=2, njmax=10)
  e.qpos[1] = -0.62   # whole body pressed into the ground: > 2 contacts
  e.forward()
  assert e.warning[mc.C['DMC_WARN_CONTACTFULL']] >= 1
  assert e.ncon[0] == 2


_ELLIPTIC_SCENE = """
<mujoco><option cone="elliptic" impratio="{impratio}" gravity="2 0.5 -9.81"/>
<default><geom friction="0.7 0.02 0.003" condim="{condim}"/></default>
<worldbody>
  <geom name='floor' type='plane' size='5 5 1' conaffinity='5'/>
  <body name='box' pos='0 0 .12'><freejoint/>
    <geom name='box' type='box' size='.1 .08 .1' contype='4' conaffinity='4'/>
    <body name='arm' pos='.1 0 .1'><joint name='h' type='hinge' axis='0 1 0' range='-60 60' limited='true'/>
      <geom name='arm' type='capsule' fromto='0 0 0 .3 0 0' size='.04'/></body></body>
  <body name='ball' pos='.5 .3 .1'><freejoint/><geom name='ball' size='.1'/></body>
</worldbody></mujoco>"""


@pytest.mark.parametrize('condim,impratio', [(3, 1.0), (4, 5.0), (6, 1.0)])
def test_elliptic_cones_match_oracle(condim, impratio):
  # cone="elliptic" (suite finger / stacker / manipulator): block rows, three-zone cost,
  # cone Hessian and the non-quadratic line search, against the oracle's restatement.
  m = mc.compile_xml(_ELLIPTIC_SCENE.format(condim=condim, impratio=impratio))
  o, e = OraclePhysics(m), EmuPhysics(m, 64)
  rs = np.random.RandomState(3)
  v = rs.uniform(-1, 1, m.nv)
  o.qvel[:] = v
  e.qvel[:] = v
  o.forward()
  e.forward()
  assert o.nefc == e.nefc[0] and o.ncon == e.ncon[0] and o.ncon > 0
  ne = o.nefc
  np.testing.assert_allclose(o.efc_J[:ne*m.nv], e.dense_J(), rtol=1e-11, atol=1e-13)
  np.testing.assert_allclose(o.efc_D[:ne], e.scratch('efc_D')[:ne], rtol=1e-12)
  np.testing.assert_allclose(o.efc_aref[:ne], e.scratch('efc_aref')[:ne], rtol=1e-9, atol=1e-9)
  np.testing.assert_allclose(o.qacc, e.qacc, rtol=1e-8, atol=1e-8)
  zones = set()
  for _ in range(300):
    o.step()
    e.step()
    zones |= set(e.scratch('efc_active')[:e.nefc[0]].tolist())
    np.testing.assert_allclose(o.qpos, e.qpos, rtol=0, atol=1e-8)
  assert zones >= {0, 1, 2}    # top, bottom and cone zones were all exercised
  np.testing.assert_allclose(o.qpos, e.qpos, rtol=0, atol=1e-9)
  assert not e.warning.any()


def test_finger_domain_matches_oracle():
  # suite finger: elliptic cones + dof friction loss (Huber rows) + framepos sensors +
  # touch on ellipsoid sites + cylinder decorations (guard test only)
  with open(os.path.join(ASSETS, 'finger.xml')) as f:
    m = mc.compile_xml(f.read())
  o, e = OraclePhysics(m), EmuPhysics(m, 64)
  o.forward()
  rs = np.random.RandomState(0)
  states, touched = set(), 0.0
  for _ in range(1000):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    states |= set(e.scratch('efc_active')[:e.nefc[0]].tolist())
    touched = max(touched, o.sensordata[14], o.sensordata[15])
  np.testing.assert_allclose(o.qpos, e.qpos, rtol=0, atol=1e-10)
  np.testing.assert_allclose(o.sensordata, e.sensordata, rtol=0, atol=1e-9)
  assert states >= {0, 1, 2} and (3 in states or 4 in states)   # cone zones and the friction row's linear zone
  assert touched > 0
  assert not e.warning.any() and not o.warning.any()


@pytest.mark.parametrize('name', ['fish', 'swimmer6', 'point_mass', 'ball_in_cup'])
def test_fluid_and_tendon_domains_match_oracle(name):
  # fish: free body in a fluid (inertia-box drag), tendon spring, position actuator on a fixed
  # tendon; swimmer: planar chain propelled by fluid forces, frame-axis sensors; point_mass:
  # motors acting through fixed tendons; ball_in_cup: length limit on a site-to-site spatial tendon
  if name.startswith('swimmer'):
    from dm_control_amd.suite import swimmer
    m = mc.compile_xml(swimmer._make_model(int(name[7:])))
  else:
    with open(os.path.join(ASSETS, name + '.xml')) as f:
      m = mc.compile_xml(f.read())
  o, e = OraclePhysics(m), EmuPhysics(m, 64)
  rs = np.random.RandomState(2)
  q = m.qpos0.copy()
  if name == 'fish':
    quat = rs.randn(4)
    q[3:7] = quat / np.linalg.norm(quat)
    q[7:] = rs.uniform(-.2, .2, m.nq - 7)
  o.qpos[:] = q
  e.qpos[:] = q
  o.forward()
  taut = 0
  for _ in range(600 if name == 'ball_in_cup' else 300):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    taut += int(e.nefc[0] > 0)
  assert np.abs(o.qpos - q).max() > 1e-3        # it moved
  if name == 'ball_in_cup':
    assert taut > 50                             # the string limit was active for a while
  np.testing.assert_allclose(o.qpos, e.qpos, rtol=0, atol=1e-11)
  np.testing.assert_allclose(o.sensordata, e.sensordata, rtol=0, atol=1e-10)
  o.forward()
  e.forward()   # the scratch dump is taken after a forward pass: compare like with like
  np.testing.assert_allclose(o.qfrc_passive, e.scratch('qfrc_passive'), rtol=1e-7, atol=1e-12)
  assert not e.warning.any()


@pytest.mark.parametrize('use_peg,insert', [(False, False), (True, False), (False, True)])
def test_manipulator_variants_match_oracle(use_peg, insert):
  # suite manipulator: elliptic cones + tendon equality (finger / thumb coupling) + a motor on a
  # fixed tendon + box touch sites; props removed per task as the reference does
  from dm_control_amd.suite import manipulator
  m = mc.compile_xml(manipulator.make_model(use_peg, insert)[0])
  assert m.neq == 1 and m.ntendon == 2
  o, e = OraclePhysics(m), EmuPhysics(m, 64)
  o.forward()
  rs = np.random.RandomState(0)
  for t in range(800):
    if t % 40 == 0:
      c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
  np.testing.assert_allclose(o.qpos, e.qpos, rtol=0, atol=1e-10)
  np.testing.assert_allclose(o.sensordata, e.sensordata, rtol=0, atol=1e-8)
  assert e.nefc[0] >= 1                      # the coupling row is always there
  assert not e.warning.any() and not o.warning.any()


def test_humanoid_cmu_rollout_fp64():
  """62 dofs, 1118 candidate pairs (ellipsoid hands against capsules, spheres and the floor): the
  fp64 scratch of this model does not fit in a CU's LDS, so the fp64 comparison of the kernel core
  with the oracle lives here; the GPU test is the fp32 teacher-forced one (test_gpu_suite.py)."""
  with open(os.path.join(ASSETS, 'humanoid_CMU.xml')) as f:
    m = mc.compile_xml(f.read())
  o, e = OraclePhysics(m), EmuPhysics(m, 64, nconmax=32)
  rs = np.random.RandomState(0)
  q = m.qpos0.copy()
  q[7:] += rs.uniform(-.3, .3, m.nq - 7)
  o.qpos[:] = q
  e.qpos[:] = q
  o.forward()
  maxcon, worst = 0, 0.0
  for _ in range(500):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    assert o.ncon == e.ncon[0]
    maxcon = max(maxcon, o.ncon)
    worst = max(worst, np.abs(e.qpos - o.qpos).max())
  assert maxcon >= 4 and o.qpos[2] < 0.5      # it fell and lies on the floor
  assert worst < 1e-9
  np.testing.assert_allclose(e.sensordata, o.sensordata, rtol=0, atol=1e-6 * max(1.0, np.abs(o.sensordata).max()))
  assert not o.warning.any() and not e.warning.any()


@pytest.mark.parametrize('cone', ['pyramidal', 'elliptic'])
@pytest.mark.parametrize('condim', [3, 4, 6])
@pytest.mark.parametrize('prec,tol', [(64, 1e-12), (32, 2e-6)])
def test_noslip_matches_oracle(cone, condim, prec, tol):
  # composer arenas run 5 noslip sweeps after the Newton solve (composer/arena.xml:4)
  xml = _ELLIPTIC_SCENE.format(condim=condim, impratio=1.0).replace('cone="elliptic"', 'cone="%s" noslip_iterations="5"' % cone)
  m = mc.compile_xml(xml)
  o, e = OraclePhysics(m), EmuPhysics(m, prec)
  rs = np.random.RandomState(3)
  v = rs.uniform(-1, 1, m.nv)
  o.qvel[:] = v
  e.qvel[:] = v
  o.forward()
  for _ in range(200):
    o.step()
    e.step()
  np.testing.assert_allclose(e.qpos, o.qpos, rtol=0, atol=tol * 200)
  np.testing.assert_allclose(e.qacc, o.qacc, rtol=0, atol=max(tol * 1e4, 1e-8) * max(1.0, np.abs(o.qacc).max()))
  assert not e.warning.any()


def test_config4_cmu_position_floor_rollout_fp64():
  """BASELINE config 4 physics (assets/cmu_2019_position_floor.xml: CMU 2019 humanoid, scaled position
  actuators with force limits, elliptic cones, 5 noslip sweeps, dt = 0.005): kernel core vs oracle in
  fp64 from the upright pose through the fall under random targets."""
  with open(os.path.join(ASSETS, 'cmu_2019_position_floor.xml')) as f:
    m = mc.compile_xml(f.read())
  assert (m.nq, m.nv, m.nu, m.nsensordata) == (63, 62, 56, 25)
  assert m.opt.noslip_iterations == 5 and m.opt.timestep == 0.005
  o, e = OraclePhysics(m), EmuPhysics(m, 64, nconmax=24)
  rs = np.random.RandomState(0)
  o.forward()
  head = m.names['body'].index('head')
  assert abs(o.xpos[3*head + 2] - 1.455) < 5e-3                         # upright (cmu_humanoid.py:174-176)
  maxcon, worst = 0, 0.0
  for i in range(360):
    c = rs.uniform(-1, 1, m.nu) * (0.3 if i < 150 else 1.0)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    maxcon = max(maxcon, o.ncon)
    worst = max(worst, np.abs(e.qpos - o.qpos).max())
  assert maxcon >= 4 and o.qpos[2] < 0.3
  assert worst < 1e-9
  np.testing.assert_allclose(e.sensordata, o.sensordata, rtol=0, atol=1e-6 * max(1.0, np.abs(o.sensordata).max()))
  assert np.abs(o.actuator_force).max() <= 150 + 1e-9                 # forcerange clamps (largest: lowerback, 150)
  assert not o.warning.any() and not e.warning.any()


@pytest.mark.parametrize('walls_and_ball', [False, True])
def test_quadruped_rollout_fp64(walls_and_ball):
  """suite quadruped (walk / run model, fetch model): filtered position servos (12 activation states),
  tendon transmissions and tendon equalities, ellipsoid torso, condim-6 priority ball, wall planes."""
  from dm_control_amd.suite import quadruped
  m = mc.compile_xml(quadruped.make_model(walls_and_ball=walls_and_ball))
  assert m.na == 12 and m.nu == 12
  o, e = OraclePhysics(m), EmuPhysics(m, 64, nconmax=24)
  rs = np.random.RandomState(0)
  q = m.qpos0.copy()
  q[2] = 0.7
  q[3:7] = rs.randn(4)
  q[3:7] /= np.linalg.norm(q[3:7])
  o.qpos[:] = q
  e.qpos[:] = q
  o.forward()
  worst, maxcon = 0.0, 0
  for _ in range(400):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    maxcon = max(maxcon, o.ncon)
    worst = max(worst, np.abs(e.qpos - o.qpos).max())
  assert maxcon >= 3 and worst < 1e-9
  np.testing.assert_allclose(e.act, o.act, rtol=0, atol=1e-12)
  assert np.abs(o.act).max() > 0.05
  assert not o.warning.any() and not e.warning.any()


def test_actuator_dynamics_closed_forms():
  # integrator: act = integral of ctrl; filter (explicit Euler): act_n = 1 - (1 - dt/tau)^n;
  # filterexact: act = 1 - exp(-t/tau); the force is gain * act + bias
  xml = """<mujoco><option timestep="0.005" gravity="0 0 0"/><worldbody>
  <body><joint name="a" type="hinge" axis="0 1 0" damping="1"/><geom type="capsule" fromto="0 0 0 .4 0 0" size=".04"/></body>
  </worldbody><actuator>
   <general name="f" joint="a" dyntype="filter" dynprm="0.1" gainprm="50" biastype="affine" biasprm="0 -50"/>
   <general name="i" joint="a" dyntype="integrator" gainprm="5" ctrllimited="true" ctrlrange="-1 0.5"/>
   <motor name="m" joint="a" gear="2"/>
   <general name="fe" joint="a" dyntype="filterexact" dynprm="0.05" gainprm="3"/>
  </actuator></mujoco>"""
  m = mc.compile_xml(xml)
  assert m.na == 3
  for P in (lambda: OraclePhysics(m, legacy_step=False), lambda: EmuPhysics(m, 64)):
    p = P()
    p.ctrl[:] = [1, 2, 0, 1]            # the integrator's ctrl is clamped to 0.5 first
    for _ in range(40):
      p.step(1, False) if isinstance(p, EmuPhysics) else p.step()
    np.testing.assert_allclose(p.act, [1 - (1 - 0.005/0.1)**40, 0.5*40*0.005, 1 - np.exp(-40*0.005/0.05)], rtol=1e-12)
    p.forward()
    np.testing.assert_allclose(p.actuator_force, [50*p.act[0] - 50*p.qpos[0], 5*p.act[1], 0, 3*p.act[2]], rtol=1e-12, atol=1e-12)
  with pytest.raises(mc.MjcfError):
    mc.compile_xml(xml.replace('timestep="0.005"', 'timestep="0.005" integrator="RK4"'))


def test_stacker_and_insert_peg_rollouts_fp64():
  """suite stacker (4 boxes) and manipulator insert_peg: box-box / capsule-box / sphere-box contacts with
  elliptic cones behind the manipulator arm; boxes dropped in a heap next to the arm, random torques."""
  from dm_control_amd.suite import stacker, manipulator
  for xml, movers in ((stacker.make_model(4)[0], ['box0', 'box1', 'box2', 'box3']), (manipulator.make_model(True, True)[0], ['peg'])):
    m = mc.compile_xml(xml)
    o, e = OraclePhysics(m), EmuPhysics(m, 64, nconmax=48)
    rs = np.random.RandomState(0)
    q = m.qpos0.copy()
    for k, name in enumerate(movers):
      jx, jz, jy = (m.jnt_qposadr[m.names['joint'].index(name + s)] for s in ('_x', '_z', '_y'))
      if name == 'peg':
        q[jx], q[jz], q[jy] = -.405, .42, 0.3      # above the slot (manipulator.xml: slot at x = -.405, z = .2)
      else:
        q[jx], q[jz], q[jy] = 0.1*rs.uniform(-1, 1), 0.1 + 0.07*k, rs.uniform(0, 6.28)
    o.qpos[:] = q
    e.qpos[:] = q
    o.forward()
    kinds, worst = set(), 0.0
    for t in range(700):
      c = rs.uniform(-1, 1, m.nu)
      o.ctrl[:] = c
      e.ctrl[:] = c
      o.step()
      e.step()
      worst = max(worst, np.abs(e.qpos - o.qpos).max())
      for i in range(o.ncon):
        ci = o.contact(i)
        kinds.add((int(m.geom_type[ci['geom1']]), int(m.geom_type[ci['geom2']])))
    assert worst < 1e-8, worst
    if len(movers) > 1:
      assert (0, 6) in kinds and (6, 6) in kinds             # boxes on the floor and on each other
    else:
      assert (3, 6) in kinds                                 # the peg's capsules against the slot boxes
    assert not o.warning.any() and not e.warning.any()


def test_config5_soccer_boxhead_rollout_fp64():
  """BASELINE config 5 physics (assets/soccer_2v2_boxhead.xml): four BoxHead walkers (root slides, steer,
  kick, rolling ball), the condim-6 priority-1 soccer ball, pitch walls and goal posts; elliptic cones + noslip.
  Players are driven into the ball and into each other."""
  with open(os.path.join(ASSETS, 'soccer_2v2_boxhead.xml')) as f:
    m = mc.compile_xml(f.read())
  assert (m.nq, m.nv, m.nu) == (31, 30, 12)
  o, e = OraclePhysics(m), EmuPhysics(m, 64, nconmax=24)
  from dm_control_amd.composer.tasks import soccer
  q = soccer.kickoff_qpos(m)
  # PyMJCF attaches every player at the origin: move the four to (-+1, +-1.5), around the ball on the centre spot
  for (qx, qy), xy in zip(soccer.addresses(m)['players'], ((-1, 1.5), (-1, -1.5), (1, 1.5), (1, -1.5))):
    q[qx], q[qy] = xy
  o.qpos[:] = q
  e.qpos[:] = q
  o.forward()
  rs = np.random.RandomState(0)
  kinds, worst = set(), 0.0
  for t in range(500):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    worst = max(worst, np.abs(e.qpos - o.qpos).max())
    for k in range(o.ncon):
      ci = o.contact(k)
      kinds.add((m.names['geom'][ci['geom1']].split('/')[-1], m.names['geom'][ci['geom2']].split('/')[-1]))
  assert worst < 1e-9, worst
  assert ('ground', 'shell') in kinds and ('ground', 'geom') in kinds     # wheels and the soccer ball on the pitch
  np.testing.assert_allclose(e.sensordata, o.sensordata, rtol=0, atol=1e-6 * max(1.0, np.abs(o.sensordata).max()))
  assert not o.warning.any() and not e.warning.any()


@pytest.mark.parametrize('prec,tol', [(64, 1e-12), (32, 2e-5)])
def test_connect_weld_joint_equalities_match_oracle(prec, tol):
  """Equality constraints between bodies (connect: 3 rows, weld: 6 rows incl. the quaternion-error rows) and
  joints (polynomial coupling), next to the tendon equalities of the suite models."""
  import test_oracle_kat as kat
  scenes = [kat._EQ_CHAIN,
            """<mujoco><option timestep="0.002"/><worldbody><geom type="plane" size="2 2 .1"/>
               <body name="box" pos=".3 .1 .7" quat=".8 .2 .4 .1"><freejoint/><geom type="box" size=".1 .05 .02" mass="2"/></body>
               <body name="b2" pos=".6 .1 .7"><freejoint/><geom type="sphere" size=".05"/></body>
               <body name="bob" pos="0 0 1"><freejoint/><geom type="sphere" size=".02" mass="1"/></body></worldbody>
               <equality><weld body1="box" body2="b2" anchor="-.15 0 0"/><connect body1="bob" anchor="0 0 .5" solref="0.004 1"/></equality></mujoco>"""]
  for xml in scenes:
    m = mc.compile_xml(xml)
    o, e = OraclePhysics(m), EmuPhysics(m, prec)
    v = np.random.RandomState(1).uniform(-.5, .5, m.nv)
    o.qvel[:] = v
    e.qvel[:] = v
    o.forward()
    e.forward()
    assert o.nefc == e.nefc[0] and o.nefc >= 9
    if prec == 64:
      ne = o.nefc
      np.testing.assert_allclose(e.dense_J(), np.array(o.efc_J[:ne*m.nv]), atol=1e-13)
    for _ in range(400):
      o.step()
      e.step()
    np.testing.assert_allclose(e.qpos, o.qpos, rtol=0, atol=tol)
    assert not o.warning.any() and not e.warning.any()


@pytest.mark.parametrize('asset', ['cheetah', 'humanoid', 'cartpole', 'quadruped'])
def test_stash_between_legacy_steps_is_bit_identical(asset):
  """The stash of the position / velocity stage (what mjData keeps between the mj_step1 that ends one legacy
  Physics.step() and the mj_step2 that begins the next, engine.py:147-162) must not change a single bit of
  the trajectory; editing the state invalidates it."""
  with open(os.path.join(ASSETS, asset + '.xml')) as f:
    m = mc.compile_xml(f.read())
  a, b = EmuPhysics(m, 64), EmuPhysics(m, 64)
  b.stash(True)
  rs = np.random.RandomState(2)
  q = m.qpos0.copy()
  if asset != 'cartpole':
    q[-4:] += rs.uniform(-.2, .2, 4)
  for e in (a, b):
    e.qpos[:] = q
  for t in range(60):
    c = rs.uniform(-1, 1, m.nu)
    for e in (a, b):
      e.ctrl[:] = c
      e.step(1 + t % 3)
    np.testing.assert_array_equal(a.qpos, b.qpos, err_msg='step %d' % t)
    np.testing.assert_array_equal(a.qvel, b.qvel)
    np.testing.assert_array_equal(a.sensordata, b.sensordata)
    np.testing.assert_array_equal(a.xpos, b.xpos)
    if t == 30:      # a host edit of the state: the stash is stale and must be ignored
      for e in (a, b):
        e.qpos[:] = q
        e.qvel[:] = 0
      b.invalidate()
  assert np.abs(a.qpos - q).max() > 1e-3


@pytest.mark.parametrize('asset,prec', [('cheetah', 64), ('cheetah', 32), ('humanoid', 64), ('cartpole', 64), ('quadruped', 64),
                                        ('cmu_2019_position_floor', 64)])
def test_kinematic_stash_is_bit_identical_and_self_validating(asset, prec):
  """The kinematic stash (poses, COM frame, velocities kept between legacy steps together with the (qpos, qvel) they
  belong to) must not change a single bit of the trajectory, and must notice by itself when the state was edited
  behind its back (writes to bound device tensors are not announced to the library)."""
  with open(os.path.join(ASSETS, asset + '.xml')) as f:
    m = mc.compile_xml(f.read())
  a, b = EmuPhysics(m, prec), EmuPhysics(m, prec)
  b.kstash(True)
  rs = np.random.RandomState(2)
  q = m.qpos0.copy()
  if asset != 'cartpole':
    q[-4:] += rs.uniform(-.2, .2, 4)
  for e in (a, b):
    e.qpos[:] = q
  for t in range(40):
    c = rs.uniform(-1, 1, m.nu)
    for e in (a, b):
      e.ctrl[:] = c
      e.step(1 + t % 3)
    for name in ('qpos', 'qvel', 'sensordata', 'xpos', 'xmat', 'geom_xpos', 'subtree_com', 'qacc', 'contact_dist'):
      np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg='%s step %d' % (name, t))
    if t == 20:      # a silent edit of the state (no invalidate): the stored (qpos, qvel) no longer match
      for e in (a, b):
        e.qpos[:] = q
        e.qvel[:] = 0.1
    if t == 30:      # velocities only
      for e in (a, b):
        e.qvel[:] *= 0.5
  assert np.abs(a.qpos - q).max() > 1e-3


@pytest.mark.parametrize('asset', ['cheetah', 'humanoid', 'hopper'])
def test_step1_step2_entry_points(asset):
  """mj_step1 and mj_step2 as separate calls (engine.py:156-162): step1; step2 is exactly mj_step, the derived
  arrays after step1 are those of the current state, and the pair tracks the oracle's step1 / step2."""
  with open(os.path.join(ASSETS, asset + '.xml')) as f:
    m = mc.compile_xml(f.read())
  a, b, o = EmuPhysics(m, 64), EmuPhysics(m, 64), OraclePhysics(m)
  b.stash(True)
  rs = np.random.RandomState(4)
  q = m.qpos0.copy()
  q[-3:] += rs.uniform(-.2, .2, 3)
  for e in (a, b, o):
    e.qpos[:] = q
  for t in range(80):
    c = rs.uniform(-1, 1, m.nu)
    for e in (a, b, o):
      e.ctrl[:] = c
    a.step(1, legacy=False)
    b.step1()
    o.step1()
    np.testing.assert_allclose(b.xpos, np.array(o.xpos), rtol=0, atol=1e-12)
    np.testing.assert_allclose(b.sensordata[:3], np.array(o.sensordata)[:3], rtol=0, atol=1e-9)
    b.step2()
    o.step2()
    np.testing.assert_array_equal(a.qpos, b.qpos, err_msg='step %d' % t)
    np.testing.assert_array_equal(a.qvel, b.qvel)
    np.testing.assert_allclose(b.qpos, np.array(o.qpos), rtol=0, atol=1e-9)
    if t == 40:      # an edit between step1 and step2: step2 recomputes the stage for the edited state
      b.step1()
      for e in (a, b):
        e.qvel[:] = 0
      b.invalidate()
      a.step(1, legacy=False)
      b.step2()
      np.testing.assert_array_equal(a.qpos, b.qpos)
      o.qpos[:] = b.qpos; o.qvel[:] = b.qvel; o.qacc_warmstart[:] = b.qacc_warmstart
  assert np.abs(a.qpos - q).max() > 1e-3


def _env_geom_rows(m, g, pos, quat, size):
  """pos(3) xmat(9) size(3) rbound(1) of one geom, computed here from the definitions (not via the library)."""
  w, x, y, z = np.asarray(quat, float) / np.linalg.norm(quat)
  mat = np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])
  t = int(m.geom_type[g])
  rb = {2: size[0], 3: size[0] + size[1], 6: float(np.linalg.norm(size))}.get(t, 0.0)
  return np.r_[pos, mat.ravel(), size, rb]


@pytest.mark.parametrize('prec,tol', [(64, 1e-10), (32, 2e-4)])
def test_per_env_world_geoms_match_single_model_oracles(prec, tol):
  """Per-environment model deltas (soccer pitch randomisation, soccer/pitch.py:612-690): walls and a goal post moved /
  resized per environment through the per-env geom table must give exactly what an oracle whose MODEL was edited
  that way gives -- the batch shares one compiled model, the reference would have recompiled."""
  m = mc.compile_xml(open(os.path.join(ASSETS, 'soccer_2v2_boxhead.xml')).read())
  from dm_control_amd.composer.tasks import soccer
  adr = soccer.addresses(m)
  bq, bv = adr['ball_q'], adr['ball_v']
  names = ['//unnamed_geom_1', '//unnamed_geom_2', '//unnamed_geom_3', '//unnamed_geom_4', 'home_goal/right_post', 'away_goal/top_post']
  wall = lambda n: n.startswith('//unnamed_geom')      # the four wall planes (soccer/pitch.py:410-420 adds them unnamed)
  ids = [m.name2id(n, 'geom') for n in names]
  rs = np.random.RandomState(0)
  for variant in range(2):
    scale = (0.3, 0.27)[variant]                       # two pitch sizes, both small enough for the ball to reach the walls
    o = OraclePhysics(m)
    om, e = o.model, EmuPhysics(m, prec, nconmax=24)
    rows = []
    for n, g in zip(names, ids):
      pos = np.array(m.geom_pos[g]) * (scale if wall(n) else 1.0) + (0 if wall(n) else rs.uniform(-1, 1, 3) * [2, 2, 0])
      size = np.array(m.geom_size[g]) * (1.0 if wall(n) else 1.5)
      quat = np.array(m.geom_quat[g])
      om.field('geom_pos')[3*g:3*g + 3] = pos
      om.field('geom_size')[3*g:3*g + 3] = size
      om.field('geom_rbound')[g] = _env_geom_rows(m, g, pos, quat, size)[15]
      rows.append(_env_geom_rows(m, g, pos, quat, size))
    e.set_env_geoms(ids, np.stack(rows))
    q = soccer.kickoff_qpos(m)
    q[bq:bq + 2] = (6.0, 3.0)
    v = np.zeros(m.nv); v[bv:bv + 3] = (40.0, 25.0, 1.0)        # a hard shot: the ball bounces off the (moved) walls
    for p in (o, e):
      p.qpos[:] = q; p.qvel[:] = v
    o.forward()
    hit_wall = False
    for t in range(300):
      c = rs.uniform(-1, 1, m.nu)
      o.ctrl[:] = c; e.ctrl[:] = c
      if prec == 32 and t:
        e.qpos[:] = o.qpos; e.qvel[:] = o.qvel; e.qacc_warmstart[:] = o.qacc_warmstart
      o.step(); e.step()
      if prec == 32 and t < 3:
        continue                     # the players' drop onto the pitch at t = 0: |qacc| ~ 1e4, a one-step fp32 transient
      np.testing.assert_allclose(e.qpos, o.qpos, rtol=0, atol=tol * max(1.0, np.abs(o.qpos).max()), err_msg='variant %d step %d' % (variant, t))
      for k in range(o.ncon):
        ci = o.contact(k)
        if wall(m.names['geom'][ci['geom1']]):
          hit_wall = True
    assert hit_wall and np.abs(o.qpos[bq:bq + 2]).max() < 40 * scale + 1.0       # the ball stayed inside the smaller pitch
    assert not o.warning.any() and not e.warning.any()


@pytest.mark.parametrize('asset,prec,tol', [('humanoid', 64, 1e-10), ('cheetah', 64, 1e-10), ('humanoid', 32, 2e-4)])
def test_xfrc_applied_matches_oracle(asset, prec, tol):
  """mjData.xfrc_applied (Cartesian force / torque at body COMs, SURVEY 8(b) state inputs): enters qfrc_smooth through
  the body Jacobians and cfrc_ext of the force / torque sensors."""
  with open(os.path.join(ASSETS, asset + '.xml')) as f:
    m = mc.compile_xml(f.read())
  o, e = OraclePhysics(m), EmuPhysics(m, prec)
  rs = np.random.RandomState(3)
  x = np.zeros((m.nbody, 6))
  for b in rs.choice(np.arange(1, m.nbody), 3, replace=False):
    x[b] = rs.uniform(-1, 1, 6) * [30, 30, 60, 3, 3, 3]
  o.xfrc_applied[:] = x.ravel()
  e.set_xfrc(x)
  free = np.zeros(m.nq); free[:] = m.qpos0
  for p in (o, e):
    p.qpos[:] = free
  o.forward()
  for t in range(150):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c; e.ctrl[:] = c
    if prec == 32 and t:
      e.qpos[:] = o.qpos; e.qvel[:] = o.qvel; e.qacc_warmstart[:] = o.qacc_warmstart
    o.step(); e.step()
    np.testing.assert_allclose(e.qpos, o.qpos, rtol=0, atol=tol * max(1, np.abs(o.qpos).max()), err_msg='step %d' % t)
  if prec == 64 and m.nsensordata:
    np.testing.assert_allclose(e.sensordata, o.sensordata, rtol=0, atol=1e-7 * max(1.0, np.abs(o.sensordata).max()))
  # and it matters: without the wrench the trajectory is another one
  o2 = OraclePhysics(m); o2.qpos[:] = free; o2.forward()
  rs = np.random.RandomState(3); rs.choice(np.arange(1, m.nbody), 3, replace=False); [rs.uniform(-1, 1, 6) for _ in range(3)]
  for t in range(150):
    o2.ctrl[:] = rs.uniform(-1, 1, m.nu); o2.step()
  assert np.abs(np.array(o2.qpos) - np.array(o.qpos)).max() > 1e-2


def _with_option(asset, **attrs):
  with open(os.path.join(ASSETS, asset + '.xml')) as f:
    xml = f.read()
  return xml.replace('<option', '<option ' + ' '.join('%s="%s"' % kv for kv in attrs.items()), 1)


@pytest.mark.parametrize('asset,prec,tol', [('cheetah', 64, 1e-10), ('cheetah', 32, 5e-4), ('humanoid', 64, 1e-9), ('hopper', 64, 1e-10)])
def test_cg_solver_matches_oracle(asset, prec, tol):
  """option solver="CG" (mj_solPrimal with flg_Newton = 0: gradient preconditioned with M^-1, Polak-Ribiere
  directions, the same line search and stopping tests): kernel core vs oracle, open loop."""
  m = mc.compile_xml(_with_option(asset, solver='CG', iterations='100'))
  assert m.opt.solver == 1
  o, e = OraclePhysics(m), EmuPhysics(m, prec)
  rs = np.random.RandomState(5)
  q = m.qpos0.copy()
  q[-3:] += rs.uniform(-0.3, 0.3, 3)
  o.qpos[:] = q
  e.qpos[:] = q
  o.forward()
  e.forward()
  iters = []
  for t in range(60):
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
    iters.append(int(e.solver_iter[0]))
    err = np.abs(o.qpos - e.qpos).max() / max(1.0, np.abs(o.qpos).max())
    assert err < tol, (t, err)
    if prec == 64:
      assert o.solver_iter == e.solver_iter[0], (t, o.solver_iter, e.solver_iter)
  assert max(iters) > 4      # CG takes more iterations than Newton ever does on these models


def test_cg_reaches_the_newton_solution():
  """Both solvers minimise the same convex cost: at a tight tolerance their accelerations agree (PGS, the dual
  solver: tests/test_pgs.py)."""
  mN = mc.compile_xml(_with_option('cheetah', iterations='200', tolerance='1e-12'))
  mC = mc.compile_xml(_with_option('cheetah', solver='CG', iterations='200', tolerance='1e-12'))
  rs = np.random.RandomState(0)
  for k in range(10):
    q = mN.qpos0.copy()
    q[1] = rs.uniform(-0.65, -0.3)
    q[2] = rs.uniform(-1, 1)
    q[3:] += rs.uniform(-0.4, 0.4, 6)
    v = rs.uniform(-2, 2, mN.nv)
    a, b, e = OraclePhysics(mN), OraclePhysics(mC), EmuPhysics(mC, 64)
    for p in (a, b, e):
      p.qpos[:], p.qvel[:], p.ctrl[:] = q, v, 0.3
      p.forward()
    scale = max(1.0, np.abs(a.qacc).max())
    assert a.nefc > 0
    assert np.abs(a.qacc - b.qacc).max() / scale < 1e-6
    assert np.abs(b.qacc - e.qacc).max() / scale < 1e-6      # at this tolerance the kernel's rounding floor on the stopping tests ends CG a few iterations earlier


@pytest.mark.parametrize('asset,nsub', [('cheetah', 1), ('humanoid', 5), ('cmu_2019_position_floor', 2)])
def test_step_with_forward_after_equals_step_then_forward(asset, nsub):
  """dmc_batch_step legacy_step 2 (the launch a composer control step ends with for observation_forward tasks): the legacy
  step followed by the rest of mj_forward at the new state in ONE pass over the position / velocity stage -- state, warm
  start, acceleration-stage sensors and every derived array equal those of a step launch followed by a forward launch."""
  from dm_control_amd.suite import common
  m = mc.compile_xml(common.read_model(asset + '.xml'))
  caps = dict(common.DEFAULT_CAPS.get(asset, {})); caps.pop('precision', None)
  a, b = EmuPhysics(m, 64, **caps), EmuPhysics(m, 64, **caps)
  rs = np.random.RandomState(0)
  q = m.qpos0.copy(); q[2] -= {'cheetah': 0.0, 'humanoid': 0.05, 'cmu_2019_position_floor': 0.25}[asset]      # feet in the floor
  for p in (a, b):
    p.qpos[:] = q
    p.forward()
  for t in range(12):
    c = rs.uniform(-1, 1, m.nu)
    a.ctrl[:] = c; b.ctrl[:] = c
    a.step(nsub, legacy=2)
    b.step(nsub); b.forward()
    for name in ('qpos', 'qvel', 'qacc', 'qacc_warmstart', 'sensordata', 'xpos', 'actuator_force'):
      np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg='%s step %d' % (name, t))
  assert a.ncon[0] > 0 or asset == 'cmu_2019_position_floor'


def test_models_without_degrees_of_freedom():
  """nv = 0 (an arena with nothing in it, a static world): the kernel core steps it -- time, poses, position sensors."""
  for xml in ('<mujoco/>', "<mujoco><worldbody><body name='p' pos='0 0 1'><geom type='sphere' size='.1'/><site name='s' pos='.1 0 0'/>"
              "</body></worldbody><sensor><framepos objtype='site' objname='s'/></sensor></mujoco>"):
    m = mc.compile_xml(xml)
    assert m.nv == 0
    o = OraclePhysics(m)
    o.forward()
    o.step(3)
    for prec in (64, 32):
      e = EmuPhysics(m, prec=prec)
      e.forward()
      e.step(3)
      assert abs(float(np.asarray(e.time).ravel()[0]) - o.time) < 1e-12
      np.testing.assert_allclose(np.asarray(e.xpos).ravel(), np.asarray(o.xpos).ravel(), atol=1e-6)
      if m.nsensordata:
        np.testing.assert_allclose(np.asarray(e.sensordata).ravel(), np.asarray(o.sensordata).ravel(), atol=1e-6)
