"""TEST INFRASTRUCTURE: runs the reference's OWN PyMJCF / composer / locomotion sources, unmodified, to obtain the XML
that `mjcf.RootElement.to_xml_string()` (dm_control/mjcf/element.py:817) produces for BASELINE configs 4 and 5.

`/root/reference` cannot be imported as a package here: lxml, absl, dm_env and the mujoco wheel are absent, and
`dm_control/mujoco/wrapper/mjbindings/{sizes,enums,constants}.py` are generated at build time.  This loader executes
the reference's source files exactly as they are on disk inside a synthetic `dm_control` package.  The seams, none of
which touches what the XML contains:

  * `lxml.etree`  -> a thin adapter over xml.etree.ElementTree (the five calls PyMJCF makes: Element, fromstring, parse,
                     tostring(pretty_print), strip_elements; Comment / PI identities);
  * `absl.flags / absl.logging` -> two flag defaults (pymjcf_debug = False) and the stdlib logger;
  * `tree` (dm-tree) -> map_structure over dict / list / tuple nests (composer/variation/variation_values.py:34);
  * `dm_env`      -> dm_control_amd.envs.dm_env_api (pure-Python restatement of the dm_env API);
  * `mujoco`, `dm_control.mujoco` -> empty stand-ins: nothing on the XML-authoring path calls into the engine (the two
                     mjlib helpers composer/entity.py uses for attachment poses, mju_mulQuat / mju_rotVecQuat, are given
                     in numpy);
  * `PIL.Image` (soccer/boxhead.py paints the jersey-number texture, a render-only asset) -> a stand-in that returns
                     blank pixels when PIL is absent;
  * `dm_control.mjcf.physics`, `export_with_assets*`, `composer.environment` (engine-bound) are not loaded.

No reference source is copied: files are read from /root/reference at call time.  scripts/make_pymjcf_goldens.py
uses this module to write tests/golden/pymjcf_*.xml; tests skip where the reference tree is absent."""
import importlib.util
import io
import logging as _pylogging
import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

from ref_root import REF  # noqa: E402  (/root/reference/dm_control, or the staged copy on the GPU box)
_SAVED = None


def available():
  return os.path.isdir(REF)


def _exec(name, path, package=False):
  spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)] if package else None)
  mod = importlib.util.module_from_spec(spec)
  sys.modules[name] = mod
  spec.loader.exec_module(mod)
  parent, _, leaf = name.rpartition('.')
  if parent and parent in sys.modules:
    setattr(sys.modules[parent], leaf, mod)
  return mod


def _stub(name, path=None):
  mod = types.ModuleType(name)
  mod.__path__ = [path] if path else []
  sys.modules[name] = mod
  parent, _, leaf = name.rpartition('.')
  if parent and parent in sys.modules:
    setattr(sys.modules[parent], leaf, mod)
  return mod


def _lxml_adapter():
  """lxml.etree as PyMJCF uses it, over xml.etree.ElementTree."""
  lx = types.ModuleType('lxml')
  lx.__path__ = []
  e = types.ModuleType('lxml.etree')
  e.Element = ET.Element
  e.SubElement = ET.SubElement
  e.Comment = ET.Comment
  e.PI = ET.ProcessingInstruction
  e.ProcessingInstruction = ET.ProcessingInstruction
  e._Element = ET.Element

  def fromstring(text, parser=None):
    if isinstance(text, str):
      text = text.encode('utf-8')
    return ET.fromstring(text)

  def parse(source, parser=None):
    return ET.parse(source)

  def tostring(element, pretty_print=False, encoding=None, **_):
    if pretty_print:
      import copy
      element = copy.deepcopy(element)
      ET.indent(element, space='  ')
    out = ET.tostring(element, encoding='unicode', short_empty_elements=True)
    if pretty_print:
      out += '\n'
    return out.encode('utf-8')

  def strip_elements(tree, *tags):
    for parent in list(tree.iter()):
      for child in list(parent):
        if '*' in tags or child.tag in tags:
          parent.remove(child)

  e.fromstring, e.parse, e.tostring, e.strip_elements = fromstring, parse, tostring, strip_elements
  e.XMLSyntaxError = ET.ParseError
  lx.etree = e
  return lx, e


def _absl_adapter():
  absl = types.ModuleType('absl')
  absl.__path__ = []
  flags = types.ModuleType('absl.flags')

  class _Flag:
    def __init__(self, default):
      self.default = default
      self.value = default

  class _Flags(dict):
    def is_parsed(self):
      return False

    def __getattr__(self, k):
      try:
        return self[k].value
      except KeyError:
        raise AttributeError(k)

  flags.FLAGS = _Flags()

  def _define(name, default, _help=None, **_):
    flags.FLAGS[name] = _Flag(default)
  for n in ('DEFINE_boolean', 'DEFINE_bool', 'DEFINE_string', 'DEFINE_integer', 'DEFINE_float', 'DEFINE_enum', 'DEFINE_list'):
    setattr(flags, n, _define)
  logging = types.ModuleType('absl.logging')
  log = _pylogging.getLogger('reference')
  for n in ('debug', 'info', 'warning', 'error', 'fatal', 'exception'):
    setattr(logging, n, getattr(log, n if n != 'fatal' else 'critical'))
  logging.warn = log.warning
  logging.log_first_n = lambda *a, **k: None
  logging.log_every_n = lambda *a, **k: None
  absl.flags, absl.logging = flags, logging
  return absl, flags, logging


def _mulquat(res, a, b):
  w1, x1, y1, z1 = a
  w2, x2, y2, z2 = b
  res = np.asarray(res)      # (the C function writes the buffer itself: no ndarray subclass hooks)
  res[:] = [w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2,
            w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2]


def _rotvecquat(res, vec, q):
  w, x, y, z = q
  R = np.array([[1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y)],
                [2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x)],
                [2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y)]])
  np.asarray(res)[:] = R @ np.asarray(vec, float)


def _quat2vel(res, quat, dt):
  """mju_quat2Vel: the angular velocity that takes the identity to `quat` in `dt` (axis * angle / dt, angle in (-pi, pi])."""
  axis = np.asarray(quat[1:4], float)
  s = np.linalg.norm(axis)
  speed = 2.0 * np.arctan2(s, quat[0])
  if speed > np.pi:
    speed -= 2.0 * np.pi
  np.asarray(res)[:] = (axis / s if s > 0 else axis) * speed / dt


def _map_structure(fn, *structs):
  """dm-tree's map_structure for the nests composer variations hold (dict / list / tuple / leaf)."""
  s0 = structs[0]
  if isinstance(s0, dict):
    return type(s0)((k, _map_structure(fn, *[s[k] for s in structs])) for k in s0)
  if isinstance(s0, (list, tuple)) and not hasattr(s0, '_fields'):
    return type(s0)(_map_structure(fn, *xs) for xs in zip(*structs))
  if hasattr(s0, '_fields'):
    return type(s0)(*[_map_structure(fn, *xs) for xs in zip(*structs)])
  return fn(*structs)


_OURS = ('dm_control', 'lxml', 'absl', 'dm_env', 'mujoco', 'tree')


def load():
  """Builds the synthetic package; returns the `dm_control` root module.  Idempotent."""
  global _SAVED
  root = sys.modules.get('dm_control')
  if root is not None and getattr(root, '_dmc_amd_pymjcf', False):
    return root
  _SAVED = {k: v for k, v in sys.modules.items() if k.split('.')[0] in _OURS}
  for k in list(_SAVED):
    del sys.modules[k]
  from dm_control_amd.envs import dm_env_api
  sys.modules['dm_env'] = dm_env_api
  sys.modules['dm_env.specs'] = dm_env_api.specs
  lx, e = _lxml_adapter()
  sys.modules['lxml'], sys.modules['lxml.etree'] = lx, e
  absl, flags, logging = _absl_adapter()
  sys.modules['absl'], sys.modules['absl.flags'], sys.modules['absl.logging'] = absl, flags, logging
  sys.modules['mujoco'] = types.ModuleType('mujoco')
  tree = types.ModuleType('tree')
  tree.map_structure = _map_structure
  sys.modules['tree'] = tree

  root = _stub('dm_control', REF)
  root._dmc_amd_pymjcf = True
  # dm_control.utils: the resource reader is two lines around open() (the real one imports absl flags for a test hook)
  _stub('dm_control.utils', os.path.join(REF, 'utils'))
  _exec('dm_control.utils.io', os.path.join(REF, 'utils/io.py'))      # pure Python: open() and os.walk()
  for n in ('transformations', 'containers', 'rewards'):
    _exec('dm_control.utils.' + n, os.path.join(REF, 'utils', n + '.py'))
  # dm_control.mujoco: stand-in (see module docstring)
  mj = _stub('dm_control.mujoco')

  class Physics:      # base class name only; never instantiated on this path
    pass
  mj.Physics = Physics
  wrapper = _stub('dm_control.mujoco.wrapper')

  class MjvOption:      # render options struct (camera observables mask geom groups in it); no physics content
    def __init__(self):
      self.geomgroup = np.ones(6, np.uint8)
      self.sitegroup = np.ones(6, np.uint8)
      self.flags = np.zeros(32, np.uint8)
  wrapper.MjvOption = MjvOption
  _exec('dm_control.mujoco.wrapper.util', os.path.join(REF, 'mujoco/wrapper/util.py'))
  mjb = _stub('dm_control.mujoco.wrapper.mjbindings')
  mjlib = types.SimpleNamespace(mju_mulQuat=_mulquat, mju_rotVecQuat=_rotvecquat, mju_quat2Vel=_quat2vel)
  mjb.mjlib = mjlib
  enums = types.ModuleType('dm_control.mujoco.wrapper.mjbindings.enums')
  sys.modules[enums.__name__] = enums
  mjb.enums = enums
  wrapper.mjbindings = mjb
  # PyMJCF proper, module by module (mjcf/__init__.py also imports physics / export_with_assets, which bind the engine)
  mjcf = _stub('dm_control.mjcf', os.path.join(REF, 'mjcf'))
  for n in ('constants', 'base', 'debugging', 'skin', 'attribute', 'schema', 'namescope', 'copier', 'element', 'parser',
            'traversal_utils'):
    _exec('dm_control.mjcf.' + n, os.path.join(REF, 'mjcf', n + '.py'))
  mjcf.Asset = mjcf.attribute.Asset
  mjcf.Element = mjcf.base.Element
  mjcf.PREFIX_SEPARATOR = mjcf.constants.PREFIX_SEPARATOR
  mjcf.RootElement = mjcf.element.RootElement
  for n in ('from_file', 'from_path', 'from_xml_string', 'from_zip'):
    setattr(mjcf, n, getattr(mjcf.parser, n))
  for n in ('commit_defaults', 'get_attachment_frame', 'get_frame_freejoint', 'get_frame_joints', 'get_freejoint'):
    setattr(mjcf, n, getattr(mjcf.traversal_utils, n))
  mjcf.Physics = Physics
  # composer without its engine-bound Environment
  comp = _stub('dm_control.composer', os.path.join(REF, 'composer'))
  _exec('dm_control.composer.constants', os.path.join(REF, 'composer/constants.py'))
  _exec('dm_control.composer.define', os.path.join(REF, 'composer/define.py'))
  _stub('dm_control.composer.variation', os.path.join(REF, 'composer/variation'))
  for n in ('base', 'variation_values'):
    _exec('dm_control.composer.variation.' + n, os.path.join(REF, 'composer/variation', n + '.py'))
  comp.variation.Variation = comp.variation.base.Variation
  comp.variation.evaluate = comp.variation.variation_values.evaluate
  for n in ('distributions', 'deterministic', 'rotations', 'noises', 'colors'):
    p = os.path.join(REF, 'composer/variation', n + '.py')
    if os.path.exists(p):
      try:
        _exec('dm_control.composer.variation.' + n, p)
      except Exception:      # pylint: disable=broad-except
        sys.modules.pop('dm_control.composer.variation.' + n, None)
  obs = _stub('dm_control.composer.observation', os.path.join(REF, 'composer/observation'))
  _exec('dm_control.composer.observation.obs_buffer', os.path.join(REF, 'composer/observation/obs_buffer.py'))
  obsv = _stub('dm_control.composer.observation.observable', os.path.join(REF, 'composer/observation/observable'))
  _exec('dm_control.composer.observation.observable.base', os.path.join(REF, 'composer/observation/observable/base.py'))
  _exec('dm_control.composer.observation.observable.mjcf', os.path.join(REF, 'composer/observation/observable/mjcf.py'))
  for n in ('Generic', 'MujocoCamera', 'MujocoFeature', 'Observable'):
    setattr(obsv, n, getattr(obsv.base, n))
  for n in ('MJCFCamera', 'MJCFFeature'):
    setattr(obsv, n, getattr(obsv.mjcf, n))
  _exec('dm_control.composer.observation.updater', os.path.join(REF, 'composer/observation/updater.py'))
  obs.Updater = obs.updater.Updater
  obs.Buffer = obs.obs_buffer.Buffer
  for n in ('entity', 'arena', 'initializer', 'robot', 'task'):
    _exec('dm_control.composer.' + n, os.path.join(REF, 'composer', n + '.py'))
  comp.Arena = comp.arena.Arena
  for n in dir(comp.constants):
    if not n.startswith('_'):
      setattr(comp, n, getattr(comp.constants, n))
  comp.cached_property, comp.observable = comp.define.cached_property, comp.define.observable
  for n in ('Entity', 'FreePropObservableMixin', 'ModelWrapperEntity', 'Observables'):
    setattr(comp, n, getattr(comp.entity, n))
  comp.Initializer = comp.initializer.Initializer
  comp.Robot = comp.robot.Robot
  comp.NullTask, comp.Task = comp.task.NullTask, comp.task.Task
  return root


def unload():
  global _SAVED
  for k in [k for k in sys.modules if k.split('.')[0] in _OURS]:
    del sys.modules[k]
  if _SAVED:
    sys.modules.update(_SAVED)
  _SAVED = None


def _load_locomotion():
  load()
  loco = sys.modules.get('dm_control.locomotion') or _stub('dm_control.locomotion', os.path.join(REF, 'locomotion'))
  return loco


def _locomotion_modules():
  _load_locomotion()
  if 'dm_control.locomotion.walkers' not in sys.modules:
    _stub('dm_control.locomotion.walkers', os.path.join(REF, 'locomotion/walkers'))
    _exec('dm_control.locomotion.walkers.initializers', os.path.join(REF, 'locomotion/walkers/initializers/__init__.py'), package=True)
    for n in ('base', 'legacy_base', 'rescale', 'scaled_actuators', 'cmu_humanoid'):
      _exec('dm_control.locomotion.walkers.' + n, os.path.join(REF, 'locomotion/walkers', n + '.py'))
  if 'dm_control.locomotion.arenas' not in sys.modules:
    _stub('dm_control.locomotion.arenas', os.path.join(REF, 'locomotion/arenas'))
    _exec('dm_control.locomotion.arenas.assets', os.path.join(REF, 'locomotion/arenas/assets/__init__.py'), package=True)
    _exec('dm_control.locomotion.arenas.floors', os.path.join(REF, 'locomotion/arenas/floors.py'))
  if 'dm_control.locomotion.tasks' not in sys.modules:
    _stub('dm_control.locomotion.tasks', os.path.join(REF, 'locomotion/tasks'))
    _exec('dm_control.locomotion.tasks.go_to_target', os.path.join(REF, 'locomotion/tasks/go_to_target.py'))
  return sys.modules['dm_control.locomotion']


def cmu2019_go_to_target():
  """locomotion/examples/basic_cmu_2019.py:97-118 (`cmu_humanoid_go_to_target`) up to the composer.Environment call:
  CMUHumanoidPositionControlled + Floor + GoToTarget(physics_timestep=0.005, control_timestep=0.03).  Returns the
  task; its model is task.root_entity.mjcf_model (what composer/environment.py:377-383 compiles)."""
  loco = _locomotion_modules()
  walker = loco.walkers.cmu_humanoid.CMUHumanoidPositionControlled()
  arena = loco.arenas.floors.Floor()
  return loco.tasks.go_to_target.GoToTarget(walker=walker, arena=arena, physics_timestep=0.005, control_timestep=0.03)


def _soccer_modules():
  """`dm_control.locomotion.soccer` with the names its __init__ (:21-44) re-exports, minus the Humanoid / Ant / mocap
  walkers (h5py and the mocap data are absent here) and the render-only camera.py."""
  loco = _locomotion_modules()
  if 'dm_control.entities' not in sys.modules:
    _stub('dm_control.entities', os.path.join(REF, 'entities'))
    props = _stub('dm_control.entities.props', os.path.join(REF, 'entities/props'))
    for n in ('position_detector', 'primitive'):
      _exec('dm_control.entities.props.' + n, os.path.join(REF, 'entities/props', n + '.py'))
    props.PositionDetector = props.position_detector.PositionDetector
    props.Primitive = props.primitive.Primitive
  if 'dm_control.locomotion.soccer' not in sys.modules:
    pkg = _stub('dm_control.locomotion.soccer', os.path.join(REF, 'locomotion/soccer'))
    for n in ('team', 'initializers', 'observables', 'soccer_ball', 'pitch', 'boxhead', 'task'):
      _exec('dm_control.locomotion.soccer.' + n, os.path.join(REF, 'locomotion/soccer', n + '.py'))
    _stub('dm_control.locomotion.soccer.camera')      # render-only (engine.MovableCamera); the name its test file imports
    for mod, names in (('boxhead', ('BoxHead',)), ('initializers', ('Initializer', 'UniformInitializer')),
                       ('observables', ('CoreObservablesAdder', 'InterceptionObservablesAdder', 'MultiObservablesAdder',
                                        'ObservablesAdder')),
                       ('pitch', ('MINI_FOOTBALL_GOAL_SIZE', 'MINI_FOOTBALL_MAX_AREA_PER_HUMANOID',
                                  'MINI_FOOTBALL_MIN_AREA_PER_HUMANOID', 'Pitch', 'RandomizedPitch')),
                       ('soccer_ball', ('regulation_soccer_ball', 'SoccerBall')), ('task', ('MultiturnTask', 'Task')),
                       ('team', ('Player', 'RGBA_BLUE', 'RGBA_RED', 'Team'))):
      for name in names:
        setattr(pkg, name, getattr(getattr(pkg, mod), name))
  return loco


def soccer_2v2_boxhead(randomizer=None):
  """locomotion/soccer/__init__.py:92-148 `load(team_size=2, walker_type=WalkerType.BOXHEAD)` up to the
  composer.Environment call (the package __init__ also imports the Ant / rodent / mocap-initialised humanoid
  walkers, which need h5py and the engine; the BoxHead path does not): SoccerBall(), four BoxHead players
  home0 / away0 / home1 / away1 (`_make_players`, :73-84), RandomizedPitch(min_size=(32, 24), max_size=(48, 36),
  keep_aspect_ratio=False, field_box=False, goal_size=None), Task(disable_walker_contacts=False).  `randomizer`
  (pitch.py:624: a callable returning the size ratio in [0, 1]; default Uniform) lets a caller fix the pitch size.  Returns
  the task."""
  loco = _soccer_modules()
  s = loco.soccer
  players = []
  for i in range(2):
    players.append((s.team.Team.HOME, s.boxhead.BoxHead(name='home%d' % i, walker_id=i, marker_rgba=s.team.RGBA_BLUE)))
    players.append((s.team.Team.AWAY, s.boxhead.BoxHead(name='away%d' % i, walker_id=i, marker_rgba=s.team.RGBA_RED)))
  players = ([s.team.Player(t, w) for t, w in players if t == s.team.Team.HOME] +
             [s.team.Player(t, w) for t, w in players if t == s.team.Team.AWAY])
  arena = s.pitch.RandomizedPitch(min_size=(32, 24), max_size=(48, 36), keep_aspect_ratio=False, field_box=False, goal_size=None,
                                  randomizer=randomizer)
  return s.task.Task(players=players, arena=arena, ball=s.soccer_ball.SoccerBall(), disable_walker_contacts=False)


# ---- engine-bound layers: dm_control.mjcf.physics (bindings) and composer.Environment on THIS package's Physics ----------
def _array_sizes(facade):
  """`mjbindings.sizes.array_sizes` for the arrays the facade serves: field -> (row size name[, columns]).  mjcf/physics.py
  (:63-119) builds the attribute table of `physics.bind(...)` from it."""
  data = {}
  rows = {'act': 'na', 'joint_q': 'nq', 'joint_v': 'nv', 'body': 'nbody', 'geom': 'ngeom', 'site': 'nsite', 'actuator': 'nu',
          'sensor': 'nsensordata', 'mocap': 'nmocap', 'joint': 'njnt', 'tendon': 'ntendon'}
  for name, (kind, ncol) in facade._FIELD_AXES.items():
    data[name] = (rows[kind],) + ((ncol,) if ncol else ())
  model = {}
  prefixes = (('body_', 'nbody'), ('jnt_', 'njnt'), ('dof_', 'nv'), ('geom_', 'ngeom'), ('site_', 'nsite'),
              ('actuator_', 'nu'), ('sensor_', 'nsensor'), ('tendon_', 'ntendon'), ('light_', 'nlight'), ('mat_', 'nmat'))
  from dm_control_amd import mjcf_compiler
  probe = mjcf_compiler.compile_xml("<mujoco><worldbody><light name='l'/><body name='b'><joint name='j'/><geom name='g' size='.1'/>"
                                    "<site name='s'/></body></worldbody><actuator><motor name='a' joint='j'/></actuator>"
                                    "<sensor><jointpos name='p' joint='j'/></sensor></mujoco>")
  for name, value in vars(probe).items():
    if isinstance(value, np.ndarray) and value.ndim in (1, 2):
      size = next((s for pre, s in prefixes if name.startswith(pre)), None)
      if name in ('qpos0', 'qpos_spring'):
        size = 'nq'
      if size:
        model[name] = (size,) + ((value.shape[1],) if value.ndim == 2 else ())
  return {'mjdata': data, 'mjmodel': model}


def bind_engine():
  """On top of load(): `dm_control.mujoco.Physics` becomes this package's facade (with view semantics: mjcf bindings keep
  the arrays they are handed), and the reference's mjcf/physics.py, rl/control.py and composer/environment.py are
  executed unmodified over it.  mjlib.mj_subtreeVel is a no-op (the backend serves subtree velocities itself)."""
  root = load()
  if getattr(root, '_dmc_amd_engine', False):
    return root
  from dm_control_amd import physics as facade
  mj = sys.modules['dm_control.mujoco']

  class Physics(facade.Physics):
    view_semantics = True
  Physics.__module__ = 'dm_control.mujoco'
  mj.Physics, mj.action_spec = Physics, facade.action_spec
  mjb = sys.modules['dm_control.mujoco.wrapper.mjbindings']
  sizes = types.ModuleType('dm_control.mujoco.wrapper.mjbindings.sizes')
  sizes.array_sizes = _array_sizes(facade)
  sys.modules[sizes.__name__] = sizes
  mjb.sizes = sizes
  mjb.mjlib.mj_subtreeVel = lambda model_ptr, data_ptr: None
  _stub('dm_control.rl', os.path.join(REF, 'rl'))
  _exec('dm_control.rl.control', os.path.join(REF, 'rl/control.py'))
  mjcf, comp = sys.modules['dm_control.mjcf'], sys.modules['dm_control.composer']
  flags = sys.modules['absl.flags']
  _exec('dm_control.mjcf.physics', os.path.join(REF, 'mjcf/physics.py'))
  mjcf.Physics = mjcf.physics.Physics
  _exec('dm_control.composer.environment', os.path.join(REF, 'composer/environment.py'))
  for n in ('Environment', 'EpisodeInitializationError', 'HOOK_NAMES', 'ObservationPadding'):
    setattr(comp, n, getattr(comp.environment, n))
  root._dmc_amd_engine = True
  del flags
  return root
