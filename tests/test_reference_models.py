"""The reference's own stand-alone MJCF files (walkers, props) through the compiler, the oracle and the kernel
core (host build).  Only runs where the reference tree is mounted (this container); skipped elsewhere --
nothing here is needed by the GPU tiers."""
import glob
import os

import numpy as np
import pytest

from dm_control_amd import mjcf_compiler as mc
from emu_lib import EmuPhysics
from oracle.oracle import OraclePhysics

from ref_root import REF  # noqa: E402  (/root/reference/dm_control, or the staged copy on the GPU box)
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')

_MODELS = ['locomotion/walkers/assets/humanoid_CMU_V2019.xml', 'locomotion/walkers/assets/humanoid_CMU_V2020.xml',
           'locomotion/walkers/assets/jumping_ball/jumping_ball_with_head.xml', 'locomotion/soccer/assets/boxhead/boxhead.xml',
           'third_party/ant/ant.xml']


@pytest.mark.parametrize('rel', _MODELS)
def test_reference_walker_files_step_identically_in_oracle_and_kernel_core(rel):
  with open(os.path.join(REF, rel)) as f:
    m = mc.compile_xml(f.read())
  assert m.nv > 0
  o, e = OraclePhysics(m), EmuPhysics(m, 64, nconmax=32)
  rs = np.random.RandomState(0)
  o.forward()
  for _ in range(30):      # stand-alone walker files have no root joint: limbs of the pinned T-pose body collide and the rollout is sensitive
    c = rs.uniform(-1, 1, m.nu)
    o.ctrl[:] = c
    e.ctrl[:] = c
    o.step()
    e.step()
  np.testing.assert_allclose(e.qpos, o.qpos, rtol=0, atol=1e-8)
  if m.na:
    np.testing.assert_allclose(e.act, o.act, rtol=0, atol=1e-12)      # V2020: filtered position actuators
  scale = max(1.0, np.abs(o.sensordata).max()) if m.nsensordata else 1.0
  np.testing.assert_allclose(e.sensordata, o.sensordata, rtol=0, atol=1e-6 * scale)


def test_every_suite_xml_except_mesh_and_hfield_models_compiles():
  common = {'./common/' + os.path.basename(p): open(p).read() for p in glob.glob(os.path.join(REF, 'suite/common/*.xml'))}
  ok, refused = [], {}
  for path in sorted(glob.glob(os.path.join(REF, 'suite/*.xml'))):
    name = os.path.basename(path)
    try:
      mc.compile_xml(open(path).read(), common)
      ok.append(name)
    except mc.MjcfError as ex:
      refused[name] = str(ex)
  assert set(refused) == {'dog.xml', 'quadruped.xml'}, refused         # meshes; height field (escape task only)
  assert all(('mesh' in v or 'hfield' in v) for v in refused.values()), refused      # dog: mesh assets / mesh geoms on moving bodies; quadruped: the escape task's height field
  assert len(ok) == 17


def test_task_registry_and_tags_match_the_reference():
  import re
  from dm_control_amd import suite
  tags = {}
  for path in glob.glob(os.path.join(REF, 'suite/*.py')):
    src = open(path).read()
    for mm in re.finditer(r"@SUITE\.add\(([^)]*)\)\s*\ndef (\w+)", src):
      tags[(os.path.basename(path)[:-3], mm.group(2))] = [t.strip(" '\"") for t in mm.group(1).split(',') if t.strip()]
  missing = set(tags) - set(suite.ALL_TASKS)
  assert all(d == 'dog' or (d, t) == ('quadruped', 'escape') for d, t in missing), missing      # meshes; height field
  assert set(suite.ALL_TASKS) <= set(tags)
  have = lambda tag: {k for k, v in tags.items() if tag in v and k in suite.ALL_TASKS}
  assert set(suite.BENCHMARKING) == have('benchmarking')
  assert set(suite.EASY) == have('easy') and set(suite.HARD) == have('hard')
  assert set(suite.EXTRA) == set(suite.ALL_TASKS) - set(suite.BENCHMARKING)
