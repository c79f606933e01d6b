"""Model-specialised kernels on demand (dm_control_amd/specialise.py, dmc_batch_attach_specialised).

CPU tier: the cache key, the generated header and a cross-compile of the plugin for an unseen model (hipcc needs no GPU).
`-m gpu`: the plugin attached to a batch gives the generic kernel's trajectory (fp64 to rounding, fp32 to the kernels'
usual distance), is refused for another model or other caps, and a baked model keeps its baked kernel."""
import os

import numpy as np
import pytest

from dm_control_amd import mjcf_compiler as mc
from dm_control_amd import specialise
from dm_control_amd.suite import common

UNSEEN = """
<mujoco>
  <option timestep="0.004"/>
  <worldbody>
    <geom name="floor" type="plane" size="3 3 .1"/>
    <body name="hip" pos="0 0 .6">
      <joint name="x" type="slide" axis="1 0 0"/><joint name="z" type="slide" axis="0 0 1"/><joint name="tilt" type="hinge" axis="0 1 0"/>
      <geom type="capsule" fromto="-.2 0 0 .2 0 0" size=".05" mass="3"/>
      <body name="leg" pos=".2 0 0">
        <joint name="knee" type="hinge" axis="0 1 0" range="-60 60" limited="true" damping=".2"/>
        <geom type="capsule" fromto="0 0 0 0 0 -.35" size=".04" mass="1"/>
        <body name="foot" pos="0 0 -.35">
          <joint name="ankle" type="hinge" axis="0 1 0" range="-40 40" limited="true" damping=".1"/>
          <geom type="capsule" fromto="-.05 0 0 .12 0 0" size=".03" mass=".4"/>
        </body>
      </body>
    </body>
  </worldbody>
  <actuator><motor joint="knee" gear="30"/><motor joint="ankle" gear="15"/></actuator>
  <sensor><jointpos joint="knee"/><subtreelinvel body="hip"/></sensor>
</mujoco>
"""


def test_key_depends_on_model_caps_precision_and_sources(tmp_path, monkeypatch):
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  m = mc.compile_xml(UNSEEN)
  k = specialise.key(m, 32, 32, (0, 0, 0))
  assert k == specialise.key(mc.compile_xml(UNSEEN), 32, 32, (0, 0, 0))
  assert k != specialise.key(m, 64, 32, (0, 0, 0)) and k != specialise.key(m, 32, 64, (0, 0, 0))
  assert k != specialise.key(m, 32, 32, (24, 0, 0))
  m2 = mc.compile_xml(UNSEEN.replace('mass="3"', 'mass="3.5"'))
  assert k != specialise.key(m2, 32, 32, (0, 0, 0))
  monkeypatch.delenv('DMC_SPECIALISE', raising=False)      # (tests/conftest.py pins 'cached' for the test tier)
  assert specialise.mode() == 'background'      # the product default: build in a background thread, switch over when done
  monkeypatch.setenv('DMC_SPECIALISE', 'cached')
  assert specialise.mode() == 'cached'
  monkeypatch.setenv('DMC_SPECIALISE', '1')
  assert specialise.mode() == 'build'
  # round 6: the offload architecture and the compiler are part of the key -- the cache is in-tree and travels
  assert 'gfx950' in specialise.toolchain_id() and 'clang version' in specialise.toolchain_id()
  monkeypatch.setattr(specialise, '_toolchain', 'gfx950|some other compiler')
  assert k != specialise.key(m, 32, 32, (0, 0, 0))
  monkeypatch.undo()
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  # round 5: the layout level of a small batch (caps[3] = jglobal + 1) and tuning flags are part of the key
  assert k != specialise.key(m, 32, 32, (0, 0, 0, 1))
  monkeypatch.setenv('DMC_SPEC_FLAGS', '-DDMC_NO_ROW_NEWBCAST')
  assert k != specialise.key(m, 32, 32, (0, 0, 0))
  monkeypatch.delenv('DMC_SPEC_FLAGS')
  assert k == specialise.key(m, 32, 32, (0, 0, 0))


def test_layout_level_override_changes_only_where_the_contact_rows_live():
  """gen_static_layouts with jlevel + 1 = 1 (caps[3]): a 17 .. 32-dof model keeps the compressed contact rows in LDS (no
  per-env global scratch) -- the layout dmc_batch_create gives a batch of at most one environment per CU; the level of a
  model that is at level 0 anyway (nv <= 16) cannot be raised by the override."""
  import subprocess, tempfile
  from dm_control_amd import build
  from dm_control_amd.suite import common
  build.generate_static_layouts()
  tool = os.path.join(build.CSRC, 'gen_static_layouts')
  def layout(name, *caps):
    m = mc.compile_xml(common.read_model(name + '.xml'))
    ints, reals = m.pack()
    with tempfile.TemporaryDirectory() as td:
      fi, fr = os.path.join(td, 'i.bin'), os.path.join(td, 'r.bin')
      ints.tofile(fi); reals.tofile(fr)
      return subprocess.check_output([tool, name, fi, fr] + [str(c) for c in caps]).decode().strip()
  assert layout('soccer_2v2_boxhead', 24, 0, 0, 1) != layout('soccer_2v2_boxhead', 24, 0, 0, 0) == layout('soccer_2v2_boxhead', 24, 0)
  assert layout('soccer_2v2_boxhead', 24, 0, 0, 2) == layout('soccer_2v2_boxhead', 24, 0)      # level 1 is its default
  assert layout('cheetah', 0, 0, 0, 2) == layout('cheetah', 0, 0)      # (an override never moves MORE out of LDS)


def test_plugin_cross_compiles_for_an_unseen_model(tmp_path, monkeypatch):
  """hipcc builds the plugin without a GPU; the object exports the three entry points the library looks up."""
  import subprocess
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  m = mc.compile_xml(UNSEEN)
  p = specialise.build(m, 32)
  assert os.path.exists(p) and p == specialise.path_for(m, 32, 32, (0, 0, 0))
  syms = subprocess.run(['nm', '-D', p], capture_output=True, text=True).stdout
  for s in ('dmc_spec_layout', 'dmc_spec_info', 'dmc_spec_launch'):
    assert s in syms
  t = os.path.getmtime(p)
  assert specialise.build(m, 32) == p and os.path.getmtime(p) == t      # cached


def _roll(b, m, T, seed=0):
  rs = np.random.RandomState(seed)
  B = b.batch_size
  q = np.tile(m.qpos0, (B, 1))
  q[:, 2:] += rs.uniform(-.3, .3, (B, m.nq - 2))
  b.set('qpos', q)
  for _ in range(T):
    b.set('ctrl', rs.uniform(-1, 1, (B, m.nu)))
    b.step()
  return b.get('qpos'), b.get('sensordata')


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol', [(64, 1e-11), (32, 2e-4)])
def test_attached_plugin_follows_the_generic_kernel(tmp_path, monkeypatch, precision, tol):
  from dm_control_amd.batch import BatchedPhysics
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  m = mc.compile_xml(UNSEEN)
  g = BatchedPhysics(m, 64, precision=precision, specialise='off')
  assert g.specialised == 'off' and g.info()['static_id'] == -1
  s = BatchedPhysics(m, 64, precision=precision, specialise='cached')
  assert s.specialised == 'missing'
  s.close()
  s = BatchedPhysics(m, 64, precision=precision, specialise='build')
  assert s.specialised == 'attached' and s.info()['static_id'] == 1000
  # fp64 open loop; fp32 step by step from the generic kernel's state (two fp32 kernels with different operation orders
  # part ways at the first just-touching contact of a 150-step rollout: what is compared is the step)
  if precision == 64:
    qg, sg = _roll(g, m, 150)
    qs, ss = _roll(s, m, 150)
    np.testing.assert_allclose(qs, qg, rtol=0, atol=tol)
    np.testing.assert_allclose(ss, sg, rtol=0, atol=tol * 50)
  else:
    rs = np.random.RandomState(0)
    q = np.tile(m.qpos0, (64, 1)); q[:, 2:] += rs.uniform(-.3, .3, (64, m.nq - 2))
    g.set('qpos', q)
    worst = 0.0
    for _ in range(150):
      c = rs.uniform(-1, 1, (64, m.nu))
      for f in ('qpos', 'qvel', 'qacc_warmstart'):
        s.set(f, g.get(f))
      for b in (g, s):
        b.set('ctrl', c); b.step()
      worst = max(worst, float(np.abs(s.get('qpos') - g.get('qpos')).max()))
    assert worst < 2e-5, worst
  assert np.isfinite(s.get('qpos')).all() and not s.get('warning').any()
  # forward / step1 / step2 / rollout modes run through the plugin too
  s.forward(); s.step1(); s.step2(); s.step(3)
  assert not s.get('warning').any()
  # a second batch of the same model finds the object in the cache
  s2 = BatchedPhysics(m, 8, precision=precision, specialise='cached')
  assert s2.specialised == 'attached'
  for b in (g, s, s2):
    b.close()


@pytest.mark.gpu
def test_plugin_of_another_model_or_other_caps_is_refused(tmp_path, monkeypatch):
  from dm_control_amd import _native
  from dm_control_amd.batch import BatchedPhysics
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  m = mc.compile_xml(UNSEEN)
  p = specialise.build(m, 32)
  other = mc.compile_xml(common.read_model('pendulum.xml'))
  b = BatchedPhysics(other, 4, precision=32, specialise='off')
  assert _native.lib().dmc_batch_attach_specialised(b._ptr, p.encode()) != 0
  assert b'layout' in _native.lib().dmc_last_error()
  b.step()
  b.close()
  b = BatchedPhysics(m, 4, precision=32, nconmax=2, specialise='off')      # same model, a contact cap below its default: another layout
  assert _native.lib().dmc_batch_attach_specialised(b._ptr, p.encode()) != 0
  assert b'layout' in _native.lib().dmc_last_error()
  b.close()
  b = BatchedPhysics(m, 4, precision=64, specialise='off')
  assert _native.lib().dmc_batch_attach_specialised(b._ptr, p.encode()) != 0
  assert b'precision' in _native.lib().dmc_last_error()
  b.close()


@pytest.mark.gpu
def test_baked_models_keep_their_baked_kernel():
  from dm_control_amd.batch import BatchedPhysics
  m = mc.compile_xml(common.read_model('cheetah.xml'))
  b = BatchedPhysics(m, 16, precision=32, specialise='build')
  assert b.specialised == 'baked' and 0 <= b.info()['static_id'] < 1000
  b.close()


@pytest.mark.gpu
def test_default_mode_builds_in_the_background_and_switches_over(tmp_path, monkeypatch, caplog):
  """The product default (DMC_SPECIALISE unset): a model never seen before starts on the generic kernel -- said so in
  one log line -- while ONE hipcc run compiles its specialised kernel in a background thread; the batch switches over at
  the first launch after the build (or at `wait_specialised()`), mid-episode, on the same state; the trajectory stays
  the generic kernel's to the kernels' usual distance and a second batch of the model finds the object in the cache."""
  import logging
  from dm_control_amd.batch import BatchedPhysics
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  monkeypatch.delenv('DMC_SPECIALISE', raising=False)
  m = mc.compile_xml(UNSEEN.replace('mass="3"', 'mass="3.25"'))
  rs = np.random.RandomState(2)
  B = 64
  q = np.tile(m.qpos0, (B, 1)); q[:, 2:] += rs.uniform(-.3, .3, (B, m.nq - 2))
  acts = rs.uniform(-1, 1, (40, B, m.nu))
  ref = BatchedPhysics(m, B, precision=64, specialise='off')
  with caplog.at_level(logging.INFO, logger='dm_control_amd.specialise'):
    b = BatchedPhysics(m, B, precision=64)
    assert b.specialised == 'building' and b.info()['static_id'] == -1
    assert any('generic step kernel' in r.getMessage() and 'background' in r.getMessage() for r in caplog.records)
    for x in (ref, b):
      x.set('qpos', q)
    for t in range(20):
      for x in (ref, b):
        x.set('ctrl', acts[t]); x.step()
    assert b.wait_specialised(timeout=600) == 'attached' and b.info()['static_id'] == 1000
    assert any('switched to its specialised' in r.getMessage() for r in caplog.records)
  for t in range(20, 40):
    for x in (ref, b):
      x.set('ctrl', acts[t]); x.step()
  np.testing.assert_allclose(b.get('qpos'), ref.get('qpos'), rtol=0, atol=1e-10)
  assert not b.get('warning').any()
  b2 = BatchedPhysics(m, B, precision=64)
  assert b2.specialised == 'attached'
  small = BatchedPhysics(m, 1, precision=32)      # (one environment behind the mujoco seam: not worth a compile)
  assert small.specialised == 'missing'
  for x in (ref, b, b2, small):
    x.close()


@pytest.mark.gpu
def test_a_refused_object_leaves_the_batch_on_the_generic_kernel(tmp_path, monkeypatch, caplog):
  """an object the library refuses (here: another model's, forced through DMC_SPEC_PLUGIN) must not raise out
  of the constructor -- the batch logs it and runs the generic kernel."""
  import logging
  from dm_control_amd.batch import BatchedPhysics
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  p = specialise.build(mc.compile_xml(UNSEEN), 32)
  monkeypatch.setenv('DMC_SPEC_PLUGIN', p)
  other = mc.compile_xml(common.read_model('pendulum.xml'))
  with caplog.at_level(logging.WARNING, logger='dm_control_amd.specialise'):
    b = BatchedPhysics(other, 32, precision=32, specialise='cached')
  assert b.specialised == 'missing' and b.info()['static_id'] == -1
  assert any('refused' in r.getMessage() for r in caplog.records)
  b.step(3)
  assert np.isfinite(b.get('qpos')).all()
  b.close()
