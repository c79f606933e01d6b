"""Kinematics pinned on the reference's MuJoCo-generated numbers: the 2 x 10 mocap frames of the CMU 2019
walker in dm_control/locomotion/mocap/test_00{1,2}.textproto store the real-MuJoCo xpos / xquat of the 30 tracking
bodies and the egocentric end effectors / appendages for given root pose and 56 joint angles.  The oracle's mj_kinematics
on the config-4 model must reproduce them to round-off."""
import os

import numpy as np
import pytest

import mocap_golden
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.suite import common
from oracle.oracle import OraclePhysics

from ref_root import REF  # noqa: E402
REF_XML = REF + '/locomotion/walkers/assets/humanoid_CMU_V2019.xml'


def _check(model, root_joint, tol):
  g = mocap_golden.load()
  q = mocap_golden.qpos_of_frames(model, g, root_joint)
  bodies = mocap_golden.tracking_bodies(model)
  assert len(bodies) == 30
  root = model.name2id('root', 'body')
  eff = [model.name2id(b, 'body') for b in g['end_effector_bodies']]
  app = [model.name2id(b, 'body') for b in g['appendage_bodies']]
  p = OraclePhysics(model)
  worst = dict(xpos=0.0, xquat=0.0, end_effectors=0.0, appendages=0.0)
  for k in range(q.shape[0]):
    p.qpos[:] = q[k]
    p.forward()
    xpos = np.array(p.xpos).reshape(-1, 3)[bodies].ravel()
    xquat = np.array(p.xquat).reshape(-1, 4)[bodies]
    gq = g['body_quaternions'][k].reshape(-1, 4)
    sign = np.sign((xquat * gq).sum(axis=1, keepdims=True))        # q and -q are the same rotation
    assert np.all(sign > 0), 'quaternion sign convention differs from MuJoCo at frame %d' % k
    worst['xpos'] = max(worst['xpos'], np.abs(xpos - g['body_positions'][k]).max())
    worst['xquat'] = max(worst['xquat'], np.abs(xquat - gq).max())
    worst['end_effectors'] = max(worst['end_effectors'], np.abs(
        mocap_golden.egocentric(p.xpos, p.xmat, root, eff) - g['end_effectors'][k]).max())
    worst['appendages'] = max(worst['appendages'], np.abs(
        mocap_golden.egocentric(p.xpos, p.xmat, root, app) - g['appendages'][k]).max())
  for k, v in worst.items():
    assert v <= tol, (k, v)
  return worst


def test_golden_fixture_shape():
  g = mocap_golden.load()
  assert g['position'].shape == (20, 3) and g['joints'].shape == (20, 56)
  assert g['body_positions'].shape == (20, 90) and g['body_quaternions'].shape == (20, 120)
  assert g['end_effectors'].shape == (20, 12) and g['appendages'].shape == (20, 15)
  assert np.abs(g['joints']).max() > 0.5            # not a rest pose only
  assert np.allclose(np.linalg.norm(g['quaternion'], axis=1), 1.0, atol=1e-12)


def test_oracle_kinematics_on_config4_model_match_mujoco_goldens():
  m = mc.compile_xml(common.read_model('cmu_2019_position_floor.xml'))
  worst = _check(m, 'walker', 1e-12)
  assert worst['xpos'] < 5e-15 and worst['xquat'] < 5e-15


@pytest.mark.skipif(not os.path.exists(REF_XML), reason='reference tree not present')
def test_oracle_kinematics_on_unmodified_reference_walker_xml():
  """The same 20 poses through the reference's own humanoid_CMU_V2019.xml (free joint added the way
  composer attaches a walker: a `freejoint` on the root body)."""
  xml = open(REF_XML).read()
  assert '<body name="root"' in xml
  xml = xml.replace('<body name="root"', '<body name="root__"', 1)
  head, tail = xml.split('<body name="root__"', 1)
  attrs, rest = tail.split('>', 1)
  xml = head + '<body name="root"' + attrs + '><freejoint name="root"/>' + rest
  m = mc.compile_xml(xml)
  _check(m, 'root', 1e-12)


def test_go_to_target_task_observables_match_mujoco_goldens_on_the_oracle_stand_in():
  """The task layer's end_effectors_pos / appendages_pos (cmu_humanoid.py:463-482) on the golden poses, CPU tier
  (oracle-backed stand-in of the device physics); the -m gpu twin is tests/test_gpu_golden.py."""
  import torch
  from composer_fake import OracleDevicePhysics
  from dm_control_amd.composer import environment
  from dm_control_amd.composer.tasks import go_to_target
  g = mocap_golden.load()
  B = g['position'].shape[0]
  task = go_to_target.GoToTarget()
  phys = OracleDevicePhysics(task.model, B, outputs=('sensordata', 'xpos', 'xmat', 'contact_geom1'))
  env = environment.Environment(task, phys, time_limit=30.0, random_state=0)
  env.reset()
  q = mocap_golden.qpos_of_frames(task.model, g, 'walker')
  phys.field('qpos').copy_(torch.from_numpy(np.ascontiguousarray(q.T)))
  phys.field('qvel').zero_()
  phys.mark_as_dirty()
  phys.forward()
  obs = task.get_observation(phys)
  np.testing.assert_allclose(obs['end_effectors_pos'].numpy(), g['end_effectors'], rtol=0, atol=1e-12)
  np.testing.assert_allclose(obs['appendages_pos'].numpy(), g['appendages'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('prec,tol', [(64, 1e-12), (32, 1e-5)])
def test_kernel_core_kinematics_match_mujoco_goldens(prec, tol):
  """The same check on the host build of the KERNEL core (tests/emu: step_core.h with one lane per environment), so the
  arithmetic that ships is pinned in the CPU tier too."""
  from emu_lib import EmuPhysics
  m = mc.compile_xml(common.read_model('cmu_2019_position_floor.xml'))
  g = mocap_golden.load()
  q = mocap_golden.qpos_of_frames(m, g, 'walker')
  bodies = mocap_golden.tracking_bodies(m)
  p = EmuPhysics(m, prec=prec, nconmax=48)
  for k in range(0, q.shape[0], 3):
    p.qpos[:] = q[k]
    p.forward()
    np.testing.assert_allclose(p.xpos.reshape(-1, 3)[bodies].ravel(), g['body_positions'][k], rtol=0, atol=tol)
    np.testing.assert_allclose(p.xquat.reshape(-1, 4)[bodies].ravel(), g['body_quaternions'][k], rtol=0, atol=tol)
