"""TEST INFRASTRUCTURE: runs the reference's OWN Python layers, unmodified, on top of this package.

`/root/reference` cannot be imported as a package here (mujoco, dm_env, absl, lxml are absent).  This loader executes
selected reference source files exactly as they are on disk -- rl/control.py, suite/base.py, suite/<domain>.py,
suite/common/__init__.py, suite/utils/randomizers.py, utils/containers.py, utils/rewards.py, utils/xml_tools.py -- inside a synthetic `dm_control` package whose
only non-reference members are the seams the survey names: `dm_env` (the pure-Python shim
dm_control_amd.envs.dm_env_api), `dm_control.mujoco` (this package's Physics facade) and the two `mjbindings` members
the suite's randomizers and the quadruped use (the mjtJoint / mjtSensor values, mju_axisAngle2Quat); `lxml.etree`, which the image lacks, is an
adapter over xml.etree.ElementTree.  No reference source is
copied: the files are read from /root/reference at test time, and the tests skip where that tree is absent."""
import importlib.util
import os
import sys
import types

from ref_root import REF  # noqa: E402  (/root/reference/dm_control, or the staged copy on the GPU box)


def available():
  return os.path.isdir(REF)


def _exec(name, path, package=False):
  spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)] if package else None)
  mod = importlib.util.module_from_spec(spec)
  sys.modules[name] = mod
  spec.loader.exec_module(mod)
  return mod


def _stub(name):
  mod = types.ModuleType(name)
  mod.__path__ = []
  sys.modules[name] = mod
  return mod


def _lxml_seam():
  """`lxml.etree` as the suite modules and utils/xml_tools.py use it, over xml.etree.ElementTree: Element / SubElement /
  XML / fromstring / parse / tostring / XMLParser, plus `getparent()` (elements remember who they were appended to)."""
  import io
  import xml.etree.ElementTree as ET

  class Elem(ET.Element):
    _parent = None

    def append(self, child):
      child._parent = self
      super().append(child)

    def insert(self, index, child):
      child._parent = self
      super().insert(index, child)

    def extend(self, children):
      for c in children:
        self.append(c)

    def remove(self, child):
      super().remove(child)
      child._parent = None

    def getparent(self):
      return self._parent

    def __deepcopy__(self, memo):
      new = Elem(self.tag, dict(self.attrib))
      new.text, new.tail = self.text, self.tail
      for c in self:
        new.append(c.__deepcopy__(memo))
      return new

  def _parser():
    return ET.XMLParser(target=ET.TreeBuilder(element_factory=Elem))

  def _strip_blank(root):
    for e in root.iter():
      if e.text is not None and not e.text.strip():
        e.text = None
      if e.tail is not None and not e.tail.strip():
        e.tail = None
    return root

  def fromstring(text, parser=None):
    p = _parser()
    p.feed(text)
    return _strip_blank(p.close())

  def parse(source, parser=None):
    text = source.read() if hasattr(source, 'read') else open(source, 'rb').read()
    return ET.ElementTree(fromstring(text))

  def SubElement(parent, tag, attrib=None, **extra):
    e = Elem(tag, dict(attrib or {}, **extra))
    parent.append(e)
    return e

  def tostring(element, pretty_print=False, encoding=None, **_):
    if isinstance(element, ET.ElementTree):
      element = element.getroot()
    if pretty_print:
      ET.indent(element, space='  ')
    return ET.tostring(element, encoding='unicode').encode('utf-8')

  lx = types.ModuleType('lxml')
  lx.__path__ = []
  e = types.ModuleType('lxml.etree')
  e.Element = lambda tag, attrib=None, **extra: Elem(tag, dict(attrib or {}, **extra))
  e.SubElement, e.fromstring, e.XML, e.parse, e.tostring = SubElement, fromstring, fromstring, parse, tostring
  e.XMLParser = lambda **kw: None
  e.XMLSyntaxError = ET.ParseError
  lx.etree = e
  sys.modules['lxml'], sys.modules['lxml.etree'] = lx, e


def _mjbindings_seam():
  """`dm_control.mujoco.wrapper.mjbindings` as the suite's randomizers use it (suite/utils/randomizers.py:19-57): the
  mjtJoint values and mju_axisAngle2Quat (quat = [cos(a/2), axis sin(a/2)], written into its first argument)."""
  import numpy as np
  wrapper = _stub('dm_control.mujoco.wrapper')
  mjb = types.ModuleType('dm_control.mujoco.wrapper.mjbindings')
  from dm_control_amd import mjcf_compiler
  C = mjcf_compiler.C      # the mjt* values of include/dmc_model_layout.h (which follow MuJoCo's enums)
  enums = types.SimpleNamespace(
      mjtJoint=types.SimpleNamespace(mjJNT_FREE=0, mjJNT_BALL=1, mjJNT_SLIDE=2, mjJNT_HINGE=3),
      mjtSensor=types.SimpleNamespace(**{'mjSENS_' + k[len('DMC_SENS_'):]: v for k, v in C.items() if k.startswith('DMC_SENS_')}))

  def mju_axisAngle2Quat(res, axis, angle):
    res[0] = np.cos(angle / 2)
    res[1:4] = np.asarray(axis, dtype=float) * np.sin(angle / 2)
  mjb.enums = enums
  mjb.mjlib = types.SimpleNamespace(mju_axisAngle2Quat=mju_axisAngle2Quat)
  sys.modules['dm_control.mujoco.wrapper.mjbindings'] = mjb
  wrapper.mjbindings = mjb
  return wrapper


def load(domain='cheetah'):
  """Returns the reference's `dm_control.suite.<domain>` module running over dm_control_amd; idempotent per domain."""
  name = 'dm_control.suite.' + domain
  if name in sys.modules and getattr(sys.modules.get('dm_control'), '_dmc_amd_shim', False):
    return sys.modules[name]
  if not getattr(sys.modules.get('dm_control'), '_dmc_amd_shim', False):
    from dm_control_amd import physics as facade
    from dm_control_amd.envs import dm_env_api
    sys.modules['dm_env'] = dm_env_api
    sys.modules['dm_env.specs'] = dm_env_api.specs
    root = _stub('dm_control')
    root._dmc_amd_shim = True
    # seam 1: dm_control.mujoco = the Physics facade of this package (engine.py's public surface)
    mj = types.ModuleType('dm_control.mujoco')
    mj.__path__ = []
    mj.Physics = facade.Physics
    mj.action_spec = facade.action_spec
    sys.modules['dm_control.mujoco'] = mj
    root.mujoco = mj
    mj.wrapper = _mjbindings_seam()
    # the reference's resource reader (utils/io.py) is two lines around open(); kept as a shim because the real one
    # imports absl flags
    utils = _stub('dm_control.utils')
    io = types.ModuleType('dm_control.utils.io')

    def GetResource(name, mode='rb'):
      with open(name, mode) as f:
        return f.read()
    io.GetResource = GetResource
    io.GetResourceAsFile = lambda name, mode='rb': open(name, mode)
    io.GetResourceFilename = lambda name: name
    sys.modules['dm_control.utils.io'] = io
    utils.io = io
    root.utils = utils
    _lxml_seam()
    utils.xml_tools = _exec('dm_control.utils.xml_tools', os.path.join(REF, 'utils/xml_tools.py'))
    utils.containers = _exec('dm_control.utils.containers', os.path.join(REF, 'utils/containers.py'))
    utils.rewards = _exec('dm_control.utils.rewards', os.path.join(REF, 'utils/rewards.py'))
    rl = _stub('dm_control.rl')
    root.rl = rl
    rl.control = _exec('dm_control.rl.control', os.path.join(REF, 'rl/control.py'))
    suite = _stub('dm_control.suite')
    root.suite = suite
    suite.common = _exec('dm_control.suite.common', os.path.join(REF, 'suite/common/__init__.py'), package=True)
    suite.base = _exec('dm_control.suite.base', os.path.join(REF, 'suite/base.py'))
    suite.utils = _exec('dm_control.suite.utils', os.path.join(REF, 'suite/utils/__init__.py'), package=True)
    suite.utils.randomizers = _exec('dm_control.suite.utils.randomizers', os.path.join(REF, 'suite/utils/randomizers.py'))
  suite = sys.modules['dm_control.suite']
  mod = _exec(name, os.path.join(REF, 'suite/%s.py' % domain))
  setattr(suite, domain, mod)
  return mod


def unload():
  for k in [k for k in sys.modules if k == 'dm_control' or k.startswith('dm_control.') or k in ('dm_env', 'dm_env.specs', 'lxml', 'lxml.etree')]:
    del sys.modules[k]
