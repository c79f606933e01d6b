"""Activation limits and complete keyframes.

* `actlimited / actrange` (mjcf/schema.xml: general / position-style actuators with a `dyntype`): mj_nextActivation
  clamps the ADVANCED activation to actrange -- compiler, oracle (against the closed-form recurrences), kernel core
  (host build) and the device path.
* Keyframes restore everything mj_resetDataKeyframe restores (engine.py:318-323 `reset(keyframe_id)`): qpos, qvel,
  ctrl and also time, act, mocap_pos, mocap_quat -- rounds 1-3 refused keyframes that carried the last four."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dm_control_amd import mjcf_compiler as mc  # noqa: E402

XML = """
<mujoco>
  <option timestep='0.004'/>
  <worldbody>
    <body name='pad' mocap='true' pos='0 0 1.5' quat='1 0 0 0'/>
    <body name='upper' pos='0 0 1'>
      <joint name='shoulder' type='hinge' axis='0 1 0' damping='0.05'/>
      <geom name='ug' type='capsule' fromto='0 0 0 .3 0 0' size='.03' mass='0.4'/>
      <body name='lower' pos='.3 0 0'>
        <joint name='elbow' type='hinge' axis='0 1 0' damping='0.05'/>
        <geom name='lg' type='capsule' fromto='0 0 0 .25 0 0' size='.025' mass='0.2'/>
        <body name='slider' pos='.25 0 0'>
          <joint name='rail' type='slide' axis='1 0 0' damping='0.5' range='-.1 .1' limited='true'/>
          <geom name='sg' type='sphere' size='.03' mass='0.1'/>
        </body>
      </body>
    </body>
  </worldbody>
  <actuator>
    <general name='integ' joint='shoulder' dyntype='integrator' gainprm='2' actlimited='true' actrange='-0.3 0.5'/>
    <general name='filt' joint='elbow' dyntype='filter' dynprm='0.05' gainprm='1.5' actrange='-0.2 0.4'/>
    <general name='exact' joint='rail' dyntype='filterexact' dynprm='0.02' gainprm='3' actlimited='true' actrange='-0.6 0.1'/>
    <motor name='plain' joint='elbow' gear='0.2'/>
  </actuator>
  <keyframe>
    <key name='zero'/>
    <key name='bent' qpos='0.4 -0.7 0.05' qvel='0.1 0 -0.2' ctrl='0.3 -0.1 0.2 0.5' time='1.25' act='0.45 -0.15 0.05'
         mpos='0.2 0.1 1.0' mquat='0 0 3 0'/>
  </keyframe>
</mujoco>
"""


def test_compiler_actlimited_and_keyframe_fields():
  m = mc.compile_xml(XML)
  assert m.na == 3 and m.nu == 4 and m.nmocap == 1
  # autolimits (the MuJoCo default): an actrange switches actlimited on for the actuator that does not say
  assert list(m.actuator_actlimited) == [1, 1, 1, 0]
  np.testing.assert_allclose(m.actuator_actrange, [[-0.3, 0.5], [-0.2, 0.4], [-0.6, 0.1], [0, 0]])
  assert m.nkey == 2
  np.testing.assert_allclose(m.key_time, [0, 1.25])
  np.testing.assert_allclose(m.key_act, [[0, 0, 0], [0.45, -0.15, 0.05]])
  # a keyframe that does not mention the mocap poses keeps the model's; quaternions are normalised
  np.testing.assert_allclose(m.key_mpos, [[0, 0, 1.5], [0.2, 0.1, 1.0]])
  np.testing.assert_allclose(m.key_mquat, [[1, 0, 0, 0], [0, 0, 1, 0]])
  ints, reals = m.pack()
  assert ints[1] == mc.C['DMC_MODEL_VERSION'] >= 12
  with pytest.raises(mc.MjcfError, match='actlimited needs a dyntype'):
    mc.compile_xml(XML.replace("<motor name='plain' joint='elbow' gear='0.2'/>",
                               "<motor name='plain' joint='elbow' actlimited='true' actrange='0 1'/>"))
  with pytest.raises(mc.MjcfError, match=r'actrange\[0\] must be < actrange\[1\]'):
    mc.compile_xml(XML.replace("actrange='-0.3 0.5'", "actrange='0.5 -0.3'"))
  with pytest.raises(mc.MjcfError, match='expected 3 numbers'):
    mc.compile_xml(XML.replace("act='0.45 -0.15 0.05'", "act='0.45 -0.15'"))


def test_oracle_activation_clamp_follows_the_recurrences():
  """act' = clip(act + dt act_dot) (integrator, filter) / clip(act + act_dot tau (1 - exp(-dt / tau))) (filterexact),
  act_dot evaluated at the unclamped ctrl: the three activations against their closed-form recurrences, well into
  saturation on both sides."""
  from oracle.oracle import OraclePhysics
  m = mc.compile_xml(XML)
  p = OraclePhysics(m)
  dt = m.opt.timestep
  act = np.zeros(3)
  rs = np.random.RandomState(3)
  hit_lo, hit_hi = np.zeros(3, bool), np.zeros(3, bool)
  for t in range(600):
    ctrl = np.where((t // 150) % 2 == 0, 1.0, -1.0) * np.array([3.0, 2.0, 1.5, 0.3]) + 0.05 * rs.randn(4)
    p.ctrl[:] = ctrl
    p.step()
    nxt = np.array([act[0] + dt * ctrl[0],
                    act[1] + dt * (ctrl[1] - act[1]) / 0.05,
                    act[2] + (ctrl[2] - act[2]) / 0.02 * 0.02 * (1 - np.exp(-dt / 0.02))])
    act = np.clip(nxt, m.actuator_actrange[:3, 0], m.actuator_actrange[:3, 1])
    np.testing.assert_allclose(p.act, act, atol=1e-15, rtol=0)
    hit_lo |= act == m.actuator_actrange[:3, 0]
    hit_hi |= act == m.actuator_actrange[:3, 1]
  assert hit_lo.all() and hit_hi.all()
  # without the flag the same model leaves the ranges
  m2 = mc.compile_xml(XML.replace("actlimited='true'", "actlimited='false'"))
  p2 = OraclePhysics(m2)
  for _ in range(300):
    p2.ctrl[:] = [1, 2, 1.5, 0]
    p2.step()
  assert p2.act[0] > 0.5 + 0.5 and p2.act[2] > 0.1 + 0.5 and p2.act[1] <= 0.4      # ('filt' is limited by autolimits)


def test_oracle_keyframe_reset_restores_time_act_and_mocap():
  from oracle.oracle import OraclePhysics
  m = mc.compile_xml(XML)
  p = OraclePhysics(m)
  for _ in range(10):
    p.ctrl[:] = 1
    p.step()
  p.reset(1)
  np.testing.assert_allclose(p.qpos, [0.4, -0.7, 0.05])
  np.testing.assert_allclose(p.qvel, [0.1, 0, -0.2])
  np.testing.assert_allclose(p.ctrl, [0.3, -0.1, 0.2, 0.5])
  np.testing.assert_allclose(p.act, [0.45, -0.15, 0.05])
  assert p.time == 1.25
  np.testing.assert_allclose(p.mocap_pos, [0.2, 0.1, 1.0])
  np.testing.assert_allclose(p.mocap_quat, [0, 0, 1, 0])
  p.forward()
  np.testing.assert_allclose(p.xpos.reshape(-1, 3)[m.name2id('pad', 'body')], [0.2, 0.1, 1.0])
  p.step()
  assert abs(p.time - (1.25 + m.opt.timestep)) < 1e-15
  p.reset(0)
  assert p.time == 0 and not p.act.any()
  np.testing.assert_allclose(p.mocap_pos, [0, 0, 1.5])
  p.reset(-1)
  assert p.time == 0 and not p.act.any() and not p.ctrl.any()


@pytest.mark.parametrize('prec,tol', [(64, 1e-11), (32, 2e-4)])
def test_kernel_core_activation_clamp_matches_oracle(prec, tol):
  from emu_lib import EmuPhysics
  from oracle.oracle import OraclePhysics
  m = mc.compile_xml(XML)
  e, o = EmuPhysics(m, prec=prec), OraclePhysics(m)
  o.forward()      # (legacy steps open with mj_step2: mjData must hold a position / velocity stage, as after Physics.reset)
  rs = np.random.RandomState(5)
  sat = 0
  for t in range(400):
    ctrl = np.where((t // 100) % 2 == 0, 1.0, -1.0) * np.array([3.0, 2.0, 1.5, 0.3]) + 0.1 * rs.randn(4)
    e.ctrl[:] = ctrl; o.ctrl[:] = ctrl
    e.step(); o.step()
    np.testing.assert_allclose(e.act, o.act, atol=tol, rtol=0)
    np.testing.assert_allclose(e.qpos, o.qpos, atol=tol * 50, rtol=0)
    sat += int(np.any(o.act == m.actuator_actrange[:3, 0]) or np.any(o.act == m.actuator_actrange[:3, 1]))
  assert sat > 100
  if prec == 64:      # a saturated activation sits on the bound exactly, in the core too
    assert np.any(np.isin(e.act, m.actuator_actrange[:3].reshape(-1)))


def test_facade_reset_keyframe_on_the_oracle_backend(oracle_backend):
  """Physics.reset(keyframe_id) (engine.py:306-323) through the facade: time, act and the mocap pose come from the
  keyframe; reset() without one goes back to mj_resetData's state."""
  from dm_control_amd import physics as physics_lib
  phys = physics_lib.Physics.from_xml_string(XML)
  phys.reset(1)
  assert phys.data.time == 1.25
  np.testing.assert_allclose(phys.data.act, [0.45, -0.15, 0.05])
  np.testing.assert_allclose(phys.data.qpos, [0.4, -0.7, 0.05])
  np.testing.assert_allclose(phys.named.data.mocap_pos['pad'], [0.2, 0.1, 1.0])
  np.testing.assert_allclose(phys.named.data.xpos['pad'], [0.2, 0.1, 1.0])
  phys.step()
  assert abs(phys.data.time - 1.254) < 1e-12
  phys.reset()
  assert phys.data.time == 0 and not np.any(phys.data.act)
  with pytest.raises(ValueError):
    phys.reset(2)
  phys.free()


@pytest.mark.gpu
@pytest.mark.parametrize('prec,tol', [(64, 1e-10), (32, 2e-4)])
def test_device_activation_clamp_and_keyframe_reset_match_oracle(prec, tol):
  """dmc_batch_reset(keyframe) for half of the batch, then per-environment controls that drive the activations into
  both bounds: act / qpos / time against one oracle per environment."""
  from dm_control_amd.batch import BatchedPhysics
  from oracle.oracle import OraclePhysics, OracleModel
  m = mc.compile_xml(XML)
  B = 10
  b = BatchedPhysics(m, B, precision=prec)
  om = OracleModel(m)
  refs = [OraclePhysics(om) for _ in range(B)]
  mask = np.zeros(B, np.uint8); mask[::2] = 1
  b.reset(mask, keyframe_id=1)
  b.forward(True)      # (Physics.reset's mj_forward with actuation disabled, as the oracles' reset() does)
  for e, o in enumerate(refs):
    o.reset(1 if mask[e] else None)      # (reset + forward: legacy steps open with mj_step2)
  np.testing.assert_allclose(b.get('time')[:, 0], np.where(mask, 1.25, 0.0))
  np.testing.assert_allclose(b.get('act')[0], [0.45, -0.15, 0.05], atol=1e-7)
  np.testing.assert_allclose(b.get('mocap_pos')[0], [0.2, 0.1, 1.0], atol=1e-7)
  np.testing.assert_allclose(b.get('mocap_pos')[1], [0, 0, 1.5])
  rs = np.random.RandomState(7)
  gain = rs.uniform(0.5, 1.5, (B, 1))
  for t in range(300):
    ctrl = np.where((t // 100) % 2 == 0, 1.0, -1.0) * gain * np.array([3.0, 2.0, 1.5, 0.3]) + 0.1 * rs.randn(B, 4)
    b.set('ctrl', ctrl)
    b.step()
    for e, o in enumerate(refs):
      o.ctrl[:] = ctrl[e]
      o.step()
  ao = np.stack([o.act for o in refs])
  np.testing.assert_allclose(b.get('act'), ao, atol=tol, rtol=0)
  np.testing.assert_allclose(b.get('qpos'), np.stack([o.qpos for o in refs]), atol=tol * 50, rtol=0)
  np.testing.assert_allclose(b.get('time')[:, 0], [o.time for o in refs], atol=1e-12)
  lo, hi = m.actuator_actrange[:3, 0], m.actuator_actrange[:3, 1]
  assert np.all(b.get('act') >= lo - 1e-7) and np.all(b.get('act') <= hi + 1e-7)
  assert np.any(np.abs(ao - lo) < 1e-12) or np.any(np.abs(ao - hi) < 1e-12)
  b.close()
