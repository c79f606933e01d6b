"""Mocap bodies (SURVEY 8(b) field list: mocap_pos / mocap_quat; mujoco/index.py:177-267 names their rows after the
bodies with body_mocapid >= 0): compiler, oracle, kernel core (host build) and -- on the GPU -- the device path."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dm_control_amd import mjcf_compiler as mc  # noqa: E402

XML = """
<mujoco>
  <option timestep='0.002'/>
  <worldbody>
    <geom name='floor' type='plane' size='5 5 .1'/>
    <body name='target' mocap='true' pos='0.3 0 0.5' quat='1 0 0 0'>
      <geom name='tgeom' type='sphere' size='.03' contype='0' conaffinity='0'/>
      <site name='tsite' pos='0 0 .1'/>
    </body>
    <body name='second' mocap='true' pos='-1 1 1' euler='0 0 90'/>
    <body name='box' pos='0 0 0.5'>
      <freejoint name='root'/>
      <geom name='bgeom' type='box' size='.05 .05 .05' mass='0.2'/>
      <body name='arm' pos='.1 0 0'>
        <joint name='hinge' type='hinge' axis='0 1 0' damping='0.01'/>
        <geom name='ageom' type='capsule' fromto='0 0 0 .15 0 0' size='.02' mass='0.05'/>
      </body>
    </body>
  </worldbody>
  <equality>
    <weld name='drag' body1='box' body2='target' solref='0.02 1'/>
  </equality>
  <sensor>
    <framepos name='tpos' objtype='site' objname='tsite'/>
  </sensor>
</mujoco>
"""


def test_compiler_assigns_mocap_ids_and_rejects_bad_mocap_bodies():
  m = mc.compile_xml(XML)
  assert m.nmocap == 2
  assert list(m.body_mocapid) == [-1, 0, 1, -1, -1]
  ints, reals = m.pack()
  assert ints[1] == mc.C['DMC_MODEL_VERSION']
  with pytest.raises(mc.MjcfError, match='child of the world'):
    mc.compile_xml("<mujoco><worldbody><body><body mocap='true'/></body></worldbody></mujoco>")
  with pytest.raises(mc.MjcfError, match='cannot have joints'):
    mc.compile_xml("<mujoco><worldbody><body mocap='true'><joint/><geom size='.1'/></body></worldbody></mujoco>")
  # a model without mocap bodies: every id is -1
  assert mc.compile_xml("<mujoco><worldbody><body><joint/><geom size='.1'/></body></worldbody></mujoco>").nmocap == 0


def test_oracle_mocap_pose_is_data_and_a_weld_follows_it():
  from oracle.oracle import OraclePhysics
  m = mc.compile_xml(XML)
  p = OraclePhysics(m)
  # mj_resetData: the mocap arrays start at the model poses
  np.testing.assert_allclose(p.mocap_pos.reshape(2, 3), [[0.3, 0, 0.5], [-1, 1, 1]])
  np.testing.assert_allclose(p.mocap_quat.reshape(2, 4)[1], [np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], atol=1e-12)
  p.forward()
  tb = m.name2id('target', 'body')
  np.testing.assert_allclose(p.xpos.reshape(-1, 3)[tb], [0.3, 0, 0.5])
  # moving the mocap body moves its frame, its geom, its site and the sensor on it; the quaternion is normalised
  p.mocap_pos[:3] = [0.1, -0.2, 0.8]
  p.mocap_quat[:4] = [2, 0, 0, 2]
  p.forward()
  np.testing.assert_allclose(p.xpos.reshape(-1, 3)[tb], [0.1, -0.2, 0.8])
  np.testing.assert_allclose(p.xquat.reshape(-1, 4)[tb], [np.sqrt(.5), 0, 0, np.sqrt(.5)], atol=1e-15)
  np.testing.assert_allclose(p.geom_xpos.reshape(-1, 3)[m.name2id('tgeom', 'geom')], [0.1, -0.2, 0.8])
  np.testing.assert_allclose(p.sensordata[:3], [0.1, -0.2, 0.9], atol=1e-15)
  # the welded box follows: the weld keeps the relative pose of the model (target = box + R_box (0.3, 0, 0), same
  # orientation), so with the target turned 90 degrees about z the box settles at target - (0, 0.3, 0), sagging a
  # little under gravity (a soft constraint)
  for _ in range(1500):
    p.step()
  bb = m.name2id('box', 'body')
  got = p.xpos.reshape(-1, 3)[bb]
  assert np.abs(got[:2] - [0.1, -0.5]).max() < 5e-3 and -0.05 < got[2] - 0.8 < 0
  np.testing.assert_allclose(np.abs(p.xquat.reshape(-1, 4)[bb]), [np.sqrt(.5), 0, 0, np.sqrt(.5)], atol=6e-2)      # (tilted a little by the arm)
  assert np.abs(p.qvel[:6]).max() < 1e-2 and p.nefc >= 6
  # the mocap body itself has no dofs and no mass in the tree
  assert m.body_dofnum[tb] == 0


@pytest.mark.parametrize('prec,tol', [(64, 1e-10), (32, 2e-3)])
def test_kernel_core_mocap_matches_oracle(prec, tol):
  from emu_lib import EmuPhysics
  from oracle.oracle import OraclePhysics
  m = mc.compile_xml(XML)
  e, o = EmuPhysics(m, prec=prec), OraclePhysics(m)
  rs = np.random.RandomState(0)
  for t in range(120):
    if t % 20 == 0:
      pos = np.array([[0.3, 0, 0.5], [-1, 1, 1]]) + rs.uniform(-0.2, 0.2, (2, 3))
      quat = rs.randn(2, 4)
      e.set_mocap(pos, quat)
      o.mocap_pos[:] = pos.reshape(-1); o.mocap_quat[:] = quat.reshape(-1)
      o.forward()      # (an edit of mjData between legacy steps: the derived arrays follow, as Physics.forward does)
    e.step(); o.step()
    np.testing.assert_allclose(e.qpos, o.qpos, atol=tol, rtol=0)
  np.testing.assert_allclose(e.xpos.reshape(-1, 3)[1], o.xpos.reshape(-1, 3)[1], atol=1e-6 if prec == 32 else 1e-14)
  np.testing.assert_allclose(e.sensordata, o.sensordata, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize('prec,tol', [(64, 1e-9), (32, 2e-3)])
def test_device_mocap_per_environment_poses_match_oracle(prec, tol):
  """Every environment of a batch drags its box to its OWN mocap target: dmc_batch fields mocap_pos / mocap_quat
  against one oracle per environment, incl. an edit in the middle of the rollout and dmc_batch_reset."""
  from dm_control_amd.batch import BatchedPhysics
  from oracle.oracle import OraclePhysics, OracleModel
  m = mc.compile_xml(XML)
  B = 12
  b = BatchedPhysics(m, B, precision=prec)
  assert b.get('mocap_pos').shape == (B, 6) and b.get('mocap_quat').shape == (B, 8)
  np.testing.assert_allclose(b.get('mocap_pos')[3], [0.3, 0, 0.5, -1, 1, 1])
  om = OracleModel(m)
  refs = [OraclePhysics(om) for _ in range(B)]
  rs = np.random.RandomState(1)
  for t in range(90):
    if t % 30 == 0:
      pos = np.tile([0.3, 0, 0.5, -1, 1, 1], (B, 1)) + rs.uniform(-0.2, 0.2, (B, 6))
      quat = rs.randn(B, 8)
      b.set('mocap_pos', pos); b.set('mocap_quat', quat)
      for e, o in enumerate(refs):
        o.mocap_pos[:] = pos[e]; o.mocap_quat[:] = quat[e]
        o.forward()
    b.step()
    for o in refs:
      o.step()
  qo = np.stack([o.qpos for o in refs])
  np.testing.assert_allclose(b.get('qpos'), qo, atol=tol, rtol=0)
  xo = np.stack([o.xpos for o in refs])
  np.testing.assert_allclose(b.get('xpos'), xo, atol=max(tol, 1e-6))
  assert np.ptp(b.get('xpos').reshape(B, -1, 3)[:, 1, 0]) > 0.05       # the targets really differ per environment
  mask = np.zeros(B, np.uint8); mask[::2] = 1
  b.reset(mask)
  mp = b.get('mocap_pos')
  np.testing.assert_allclose(mp[0], [0.3, 0, 0.5, -1, 1, 1])
  assert np.abs(mp[1] - [0.3, 0, 0.5, -1, 1, 1]).max() > 1e-3
  b.close()


@pytest.mark.gpu
def test_facade_mocap_arrays_and_named_rows():
  """physics.data.mocap_pos / mocap_quat and their named rows (mujoco/index.py:177-267: rows named after the mocap
  bodies), the mjtState components, and a written pose reaching the kinematics."""
  from dm_control_amd import physics as physics_lib
  phys = physics_lib.Physics.from_xml_string(XML)
  assert phys.data.mocap_pos.shape == (2, 3) and phys.data.mocap_quat.shape == (2, 4)
  np.testing.assert_allclose(phys.named.data.mocap_pos['second'], [-1, 1, 1])
  phys.named.data.mocap_pos['target'] = [0.0, 0.4, 0.7]
  phys.named.data.mocap_quat['target'] = [0, 1, 0, 0]
  phys.forward()
  np.testing.assert_allclose(phys.named.data.xpos['target'], [0.0, 0.4, 0.7])
  np.testing.assert_allclose(phys.named.data.xquat['target'], [0, 1, 0, 0])
  np.testing.assert_allclose(phys.named.data.site_xpos['tsite'], [0.0, 0.4, 0.6], atol=1e-12)
  with pytest.raises(KeyError):
    phys.named.data.mocap_pos['box']
  # mjSTATE_MOCAP_POS | mjSTATE_MOCAP_QUAT
  s = phys.get_state((1 << 9) | (1 << 10))
  np.testing.assert_allclose(s, np.concatenate([[0.0, 0.4, 0.7, -1, 1, 1], phys.data.mocap_quat.reshape(-1)]))
  s2 = s.copy(); s2[:3] = [1, 2, 3]
  phys.set_state(s2, (1 << 9) | (1 << 10))
  phys.forward()
  np.testing.assert_allclose(phys.named.data.xpos['target'], [1, 2, 3])
  phys.free()
