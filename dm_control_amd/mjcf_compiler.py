"""MJCF -> flat constant tables ("compiled model").

The reference delegates this to MuJoCo's (third-party, absent) XML compiler via
`wrapper.MjModel.from_xml_string` (dm_control/mujoco/wrapper/core.py:151-182,
289).  This module re-implements the subset of MJCF semantics that the suite /
locomotion models on the hot path use: includes,
nested default classes + childclass, compiler angle/settotalmass, bodies with
every orientation spec, joints (free/ball/slide/hinge), primitive geoms with
fromto, inertia-from-geoms, sites, motor/position/velocity/general actuators,
sensors, contact excludes, keyframes, and the compile-time constants MuJoCo
derives at qpos0 (body_invweight0, dof_invweight0, stat.meaninertia).

Default attribute values follow dm_control/mjcf/schema.xml (the reference's
own copy of MuJoCo's schema, e.g. option :51-80, geom :311-347, joint :285-310).

Output: `Model` with numpy arrays named like mjModel fields, plus `pack()` which
serialises them into the blob described by include/dmc_model_layout.h.
"""
import collections
import threading
import copy
import hashlib
import contextlib
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from dm_control_amd import _layout

C = _layout.CONSTS
MINVAL = C['DMC_MINVAL']

_JNT = {'free': 0, 'ball': 1, 'slide': 2, 'hinge': 3}
_GEOM = {'plane': 0, 'hfield': 1, 'sphere': 2, 'capsule': 3, 'ellipsoid': 4,
         'cylinder': 5, 'box': 6, 'mesh': 7}
_DISABLE_FLAGS = ['constraint', 'equality', 'frictionloss', 'limit', 'contact',
                  'spring', 'damper', 'gravity', 'clampctrl', 'warmstart',
                  'filterparent', 'actuation', 'refsafe', 'sensor', 'midphase',
                  'eulerdamp', 'autoreset', 'nativeccd', 'island', 'multiccd']
_ENABLE_FLAGS = ['override', 'energy', 'fwdinv', 'invdiscrete', 'sleep',
                 'diagexact']
_ACTUATOR_TAGS = ('motor', 'position', 'velocity', 'general', 'cylinder')
# sensor tag -> (type, object attribute, objtype, dim, needstage)
_SENSORS = {
    'touch': (C['DMC_SENS_TOUCH'], 'site', C['DMC_OBJ_SITE'], 1, 3),
    'accelerometer': (C['DMC_SENS_ACCELEROMETER'], 'site', C['DMC_OBJ_SITE'], 3, 3),
    'velocimeter': (C['DMC_SENS_VELOCIMETER'], 'site', C['DMC_OBJ_SITE'], 3, 2),
    'gyro': (C['DMC_SENS_GYRO'], 'site', C['DMC_OBJ_SITE'], 3, 2),
    'force': (C['DMC_SENS_FORCE'], 'site', C['DMC_OBJ_SITE'], 3, 3),
    'torque': (C['DMC_SENS_TORQUE'], 'site', C['DMC_OBJ_SITE'], 3, 3),
    'jointpos': (C['DMC_SENS_JOINTPOS'], 'joint', C['DMC_OBJ_JOINT'], 1, 1),
    'jointvel': (C['DMC_SENS_JOINTVEL'], 'joint', C['DMC_OBJ_JOINT'], 1, 2),
    'actuatorfrc': (C['DMC_SENS_ACTUATORFRC'], 'actuator', C['DMC_OBJ_ACTUATOR'], 1, 3),
    'subtreecom': (C['DMC_SENS_SUBTREECOM'], 'body', C['DMC_OBJ_BODY'], 3, 1),
    'subtreelinvel': (C['DMC_SENS_SUBTREELINVEL'], 'body', C['DMC_OBJ_BODY'], 3, 2),
    # object named by objtype/objname (body = inertial frame, xbody = body frame, geom, site)
    'framepos': (C['DMC_SENS_FRAMEPOS'], 'objname', None, 3, 1),
    'framexaxis': (C['DMC_SENS_FRAMEXAXIS'], 'objname', None, 3, 1),
    'frameyaxis': (C['DMC_SENS_FRAMEYAXIS'], 'objname', None, 3, 1),
    'framezaxis': (C['DMC_SENS_FRAMEZAXIS'], 'objname', None, 3, 1),
    'rangefinder': (C['DMC_SENS_RANGEFINDER'], 'site', C['DMC_OBJ_SITE'], 1, 1),
    'framequat': (C['DMC_SENS_FRAMEQUAT'], 'objname', None, 4, 1),
    'framelinvel': (C['DMC_SENS_FRAMELINVEL'], 'objname', None, 3, 2),
    'frameangvel': (C['DMC_SENS_FRAMEANGVEL'], 'objname', None, 3, 2),
}


class MjcfError(ValueError):
  """Model-load failure; the reference surfaces these as ValueError
  (dm_control/mujoco/wrapper/core_test.py:71-72,96-98)."""


# ----------------------------------------------------------------------------
# small math helpers (quaternions are wxyz, Hamilton product, v' = q v q*; the
# conventions the reference pins in dm_control/utils/transformations.py)
# ----------------------------------------------------------------------------
def _vec(s, n=None):
  v = np.array([float(x) for x in s.split()], dtype=np.float64)
  if n is not None and v.size != n:
    raise MjcfError('expected %d numbers, got %r' % (n, s))
  return v


def _rgba(a):
  v = _vec(a.get('rgba', '0.5 0.5 0.5 1'))
  return v if v.size == 4 else np.array([0.5, 0.5, 0.5, 1.0])


def quat_mul(a, b):
  return np.array([
      a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
      a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
      a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
      a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]])


def quat_to_mat(q):
  w, x, y, z = q
  return np.array([
      [w*w + x*x - y*y - z*z, 2*(x*y - w*z), 2*(x*z + w*y)],
      [2*(x*y + w*z), w*w - x*x + y*y - z*z, 2*(y*z - w*x)],
      [2*(x*z - w*y), 2*(y*z + w*x), w*w - x*x - y*y + z*z]])


def mat_to_quat(m):
  """Rotation matrix -> unit quaternion (w >= 0 branch by largest diagonal)."""
  t = np.trace(m)
  if t > 0:
    s = math.sqrt(t + 1.0) * 2
    q = np.array([0.25*s, (m[2, 1]-m[1, 2])/s, (m[0, 2]-m[2, 0])/s, (m[1, 0]-m[0, 1])/s])
  elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
    s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
    q = np.array([(m[2, 1]-m[1, 2])/s, 0.25*s, (m[0, 1]+m[1, 0])/s, (m[0, 2]+m[2, 0])/s])
  elif m[1, 1] > m[2, 2]:
    s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
    q = np.array([(m[0, 2]-m[2, 0])/s, (m[0, 1]+m[1, 0])/s, 0.25*s, (m[1, 2]+m[2, 1])/s])
  else:
    s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
    q = np.array([(m[1, 0]-m[0, 1])/s, (m[0, 2]+m[2, 0])/s, (m[1, 2]+m[2, 1])/s, 0.25*s])
  return q / np.linalg.norm(q)


def axisangle_to_quat(axis, angle):
  axis = np.asarray(axis, dtype=np.float64)
  n = np.linalg.norm(axis)
  if n < MINVAL or angle == 0:
    return np.array([1.0, 0, 0, 0])
  s = math.sin(angle / 2)
  return np.concatenate([[math.cos(angle / 2)], axis / n * s])


def z_to_quat(vec):
  """Minimal rotation taking +z to `vec` (MJCF zaxis / fromto semantics)."""
  vec = np.asarray(vec, dtype=np.float64)
  n = np.linalg.norm(vec)
  if n < MINVAL:
    return np.array([1.0, 0, 0, 0])
  vec = vec / n
  axis = np.cross([0.0, 0, 1], vec)
  s = np.linalg.norm(axis)
  if s < 1e-10:
    if vec[2] > 0:
      return np.array([1.0, 0, 0, 0])
    return np.array([0.0, 1.0, 0, 0])  # 180 deg about x
  ang = math.atan2(s, vec[2])
  return axisangle_to_quat(axis / s, ang)


def rot_vec(q, v):
  return quat_to_mat(q) @ np.asarray(v, dtype=np.float64)


# ----------------------------------------------------------------------------
# defaults
# ----------------------------------------------------------------------------
class _DefaultClass:

  def __init__(self, name, parent):
    self.name = name
    self.parent = parent
    self.attrs = {}  # tag -> dict
    if parent is not None:
      for k, v in parent.attrs.items():
        self.attrs[k] = dict(v)

  def get(self, tag):
    return self.attrs.get(tag, {})


def _actuator_to_general(tag, a):
  """Rewrites motor/position/velocity shortcut attributes as `general` ones
  (schema: dm_control/mjcf/schema.xml:447-476 and the shortcut elements)."""
  out = {k: v for k, v in a.items() if k not in ('kp', 'kv', 'timeconst', 'area', 'diameter', 'bias')}
  if tag == 'cylinder':
    # pneumatic / hydraulic cylinder (schema.xml <cylinder>): first-order filter on the control with time constant
    # `timeconst`, gain = piston area (from `diameter` when given), affine bias `bias`
    area = float(a.get('area', 1))
    if 'diameter' in a:
      area = np.pi / 4 * float(a['diameter']) ** 2
    out['dyntype'] = 'filter'
    out['dynprm'] = '%r' % float(a.get('timeconst', 1))
    out['gaintype'] = 'fixed'
    out['gainprm'] = '%r' % area
    out['biastype'] = 'affine'
    out['biasprm'] = a.get('bias', '0 0 0')
  elif tag == 'motor':
    out.setdefault('gaintype', 'fixed')
    out.setdefault('biastype', 'none')
    out.setdefault('dyntype', 'none')
    out.setdefault('gainprm', '1')
  elif tag == 'position':
    kp = float(a.get('kp', 1))
    kv = float(a.get('kv', 0))
    out['gaintype'] = 'fixed'
    out['biastype'] = 'affine'
    out.setdefault('dyntype', 'none')
    out['gainprm'] = '%r' % kp
    out['biasprm'] = '0 %r %r' % (-kp, -kv)
  elif tag == 'velocity':
    kv = float(a.get('kv', 1))
    out['gaintype'] = 'fixed'
    out['biastype'] = 'affine'
    out.setdefault('dyntype', 'none')
    out['gainprm'] = '%r' % kv
    out['biasprm'] = '0 0 %r' % (-kv)
  return out


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------
def _stl_vertices(data):
  """Vertices (V, 3) of an STL asset: binary (80-byte header, uint32 triangle count, 50 bytes per facet) or ASCII."""
  if len(data) >= 84:
    n = int(np.frombuffer(data[80:84], dtype='<u4')[0])
    if len(data) == 84 + 50 * n:
      rec = np.frombuffer(data[84:], dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]))
      return rec['v'].reshape(-1, 3).astype(np.float64)
  verts = [[float(x) for x in line.split()[1:4]] for line in data.decode('latin-1').splitlines()
           if line.strip().startswith('vertex')]
  if not verts:
    raise MjcfError('mesh asset is not an STL file')
  return np.array(verts, dtype=np.float64)


def _geom_volume_inertia(gtype, size):
  """Volume and unit-density principal inertia (about the geom centre, in the
  geom frame) of a primitive.  Capsule = cylinder + two hemispheres."""
  if gtype == _GEOM['sphere']:
    r = size[0]
    v = 4.0 / 3.0 * math.pi * r**3
    i = 0.4 * v * r * r
    return v, np.array([i, i, i])
  if gtype == _GEOM['capsule']:
    r, h = size[0], 2 * size[1]
    vc = math.pi * r * r * h
    vs = 4.0 / 3.0 * math.pi * r**3
    ixy = vc * (3 * r * r + h * h) / 12.0
    iz = vc * r * r / 2.0
    si = 0.4 * vs * r * r
    ixy += si + vs * h * (3 * r + 2 * h) / 8.0
    iz += si
    return vc + vs, np.array([ixy, ixy, iz])
  if gtype == _GEOM['cylinder']:
    r, h = size[0], 2 * size[1]
    v = math.pi * r * r * h
    return v, np.array([v * (3*r*r + h*h) / 12.0, v * (3*r*r + h*h) / 12.0, v*r*r/2.0])
  if gtype == _GEOM['box']:
    a, b, c = size
    v = 8 * a * b * c
    return v, np.array([v*(b*b + c*c)/3.0, v*(a*a + c*c)/3.0, v*(a*a + b*b)/3.0])
  if gtype == _GEOM['ellipsoid']:
    a, b, c = size
    v = 4.0 / 3.0 * math.pi * a * b * c
    return v, np.array([v*(b*b + c*c)/5.0, v*(a*a + c*c)/5.0, v*(a*a + b*b)/5.0])
  return 0.0, np.zeros(3)


def _geom_rbound(gtype, size):
  if gtype == _GEOM['sphere']:
    return size[0]
  if gtype == _GEOM['capsule']:
    return size[0] + size[1]
  if gtype == _GEOM['cylinder']:
    return math.sqrt(size[0]**2 + size[1]**2)
  if gtype == _GEOM['box']:
    return float(np.linalg.norm(size))
  if gtype == _GEOM['ellipsoid']:
    return float(max(size))
  return 0.0


class _Opt:
  """model.opt namespace (subset of mjOption read by the reference:
  timestep/integrator/disableflags/gravity, dm_control/mujoco/engine.py:154,
  wrapper/core.py:389-426)."""

  def __init__(self):
    self.timestep = 0.002
    self.gravity = np.array([0.0, 0.0, -9.81])
    self.impratio = 1.0
    self.density = 0.0
    self.viscosity = 0.0
    self.tolerance = 1e-8
    self.ls_tolerance = 0.01
    self.noslip_tolerance = 1e-6
    self.integrator = 0
    self.cone = 0
    self.solver = 2
    self.iterations = 100
    self.ls_iterations = 50
    self.noslip_iterations = 0
    self.disableflags = 0
    self.enableflags = 0


class Model:
  """Compiled constant tables; attribute names follow mjModel."""

  def __init__(self):
    self.opt = _Opt()
    self.names = {}      # objtype string -> list of names (index = id)
    self.stat_meaninertia = 1.0
    self.model_name = ''

  @property
  def ptr(self):
    """MjModel.ptr (wrapper/core.py: the raw mujoco.MjModel that engine calls take): there is no such object here; the
    compiled model stands in as its own opaque handle, so code that only passes it on keeps working."""
    return self

  @property
  def name(self):
    """MjModel.name (wrapper/core.py:428-431): the `model` attribute of <mujoco>."""
    return self.model_name

  # --- reference API: MjModel.disable (wrapper/core.py:389-426) ---
  @contextlib.contextmanager
  def disable(self, *flags):
    """Temporarily sets mjtDisableBit flags, by lower-case name ('gravity', 'contact', 'actuation', ...) or value."""
    bits = {k[len('DMC_DSBL_'):].lower(): v for k, v in C.items() if k.startswith('DMC_DSBL_')}
    old = self.opt.disableflags
    new = old
    for flag in flags:
      if isinstance(flag, str):
        if flag not in bits:
          raise ValueError("'{}' is not a valid flag name. Valid names: {}".format(flag, ', '.join(bits)))
        flag = bits[flag]
      elif int(flag) not in bits.values():
        raise ValueError('{!r} is not a valid mjtDisableBit'.format(flag))
      new |= int(flag)
    self.opt.disableflags = new
    try:
      yield
    finally:
      self.opt.disableflags = old

  # --- reference API: MjModel.name2id / id2name (wrapper/core.py:347-387) ---
  def name2id(self, name, object_type):
    lst = self.names.get(object_type)
    if lst is None or name not in lst:
      raise ValueError('No %s with name %r exists.' % (object_type, name))
    return lst.index(name)

  def id2name(self, object_id, object_type):
    lst = self.names.get(object_type, [])
    if not 0 <= object_id < len(lst):
      raise ValueError('%s id %d out of range' % (object_type, object_id))
    return lst[object_id] or ''

  def sizes(self):
    return {k: int(getattr(self, k)) for k in
            ('nq', 'nv', 'nu', 'na', 'nbody', 'njnt', 'ngeom', 'nsite',
             'nsensor', 'nsensordata', 'npair', 'nkey', 'ntendon', 'nwrap', 'neq', 'nmocap')}

  def pack(self):
    """Serialises into (ints int32[], reals float64[]) per dmc_model_layout.h."""
    sizes = self.sizes()
    hdr = dict(sizes)
    o = self.opt
    hdr.update(opt_integrator=o.integrator, opt_cone=o.cone, opt_solver=o.solver,
               opt_iterations=o.iterations, opt_ls_iterations=o.ls_iterations,
               opt_noslip_iterations=o.noslip_iterations,
               opt_disableflags=o.disableflags, opt_enableflags=o.enableflags)
    ints = [C['DMC_MODEL_MAGIC'], C['DMC_MODEL_VERSION']]
    ints += [int(hdr[k]) for k in _layout.HEADER_INTS]
    rh = dict(opt_timestep=o.timestep, opt_gravity_x=o.gravity[0],
              opt_gravity_y=o.gravity[1], opt_gravity_z=o.gravity[2],
              opt_impratio=o.impratio, opt_tolerance=o.tolerance,
              opt_ls_tolerance=o.ls_tolerance,
              opt_noslip_tolerance=o.noslip_tolerance,
              opt_density=o.density, opt_viscosity=o.viscosity,
              stat_meaninertia=self.stat_meaninertia)
    reals = [float(rh[k]) for k in _layout.HEADER_REALS]
    ints = [np.asarray(ints, dtype=np.int64)]
    reals = [np.asarray(reals, dtype=np.float64)]
    for name, expr in _layout.INT_FIELDS:
      n = _layout.field_count(expr, sizes)
      a = np.asarray(getattr(self, name), dtype=np.int64).reshape(-1)
      if a.size != n:
        raise MjcfError('field %s has %d entries, expected %d' % (name, a.size, n))
      ints.append(a)
    for name, expr in _layout.REAL_FIELDS:
      n = _layout.field_count(expr, sizes)
      a = np.asarray(getattr(self, name), dtype=np.float64).reshape(-1)
      if a.size != n:
        raise MjcfError('field %s has %d entries, expected %d' % (name, a.size, n))
      reals.append(a)
    return (np.ascontiguousarray(np.concatenate(ints).astype(np.int32)),
            np.ascontiguousarray(np.concatenate(reals)))


# ----------------------------------------------------------------------------
# the compiler
# ----------------------------------------------------------------------------
class _Compiler:

  def __init__(self, xml_string, assets):
    self.assets = assets or {}
    try:
      self.root = ET.fromstring(xml_string)
    except ET.ParseError as e:
      raise MjcfError('XML parse error: %s' % e)
    if self.root.tag != 'mujoco':
      raise MjcfError('root element must be <mujoco>')
    self._expand_includes(self.root)
    # compiler settings (schema.xml:10-32)
    self.angle_deg = True
    self.eulerseq = 'xyz'
    self.settotalmass = -1.0
    self.boundmass = 0.0
    self.boundinertia = 0.0
    self.autolimits = True
    self.inertiafromgeom = 'auto'
    self.inertiagrouprange = (0, 5)
    # accumulators
    self.bodies = []   # dicts
    self.joints = []
    self.geoms = []
    self.sites = []
    self.lights = []
    self.cameras = []
    self.camera_specs = []
    self.meshes = {}      # name -> (V, 3) vertices of the STL asset
    self.actuators = []
    self.sensors = []
    self.excludes = []
    self.keys = []

  # -- includes ---------------------------------------------------------------
  def _asset_text(self, fname):
    cands = [fname, fname.lstrip('./'), './' + fname.lstrip('./')]
    for c in cands:
      if c in self.assets:
        t = self.assets[c]
        return t.decode() if isinstance(t, bytes) else t
    base = fname.split('/')[-1]
    for k, t in self.assets.items():
      if k.split('/')[-1] == base:
        return t.decode() if isinstance(t, bytes) else t
    try:
      with open(fname) as f:
        return f.read()
    except OSError:
      raise MjcfError('include file %r not found in assets' % fname)

  def _asset_bytes(self, fname):
    for c in (fname, fname.lstrip('./'), './' + fname.lstrip('./')):
      if c in self.assets:
        t = self.assets[c]
        return t if isinstance(t, bytes) else t.encode('latin-1')
    base = fname.split('/')[-1]
    for k, t in self.assets.items():
      if k.split('/')[-1] == base:
        return t if isinstance(t, bytes) else t.encode('latin-1')
    try:
      with open(fname, 'rb') as f:
        return f.read()
    except OSError:
      raise MjcfError('mesh / texture asset file %r not found' % fname)

  def _expand_includes(self, elem):
    i = 0
    children = list(elem)
    for child in children:
      if child.tag == 'include':
        sub = ET.fromstring(self._asset_text(child.attrib['file']))
        self._expand_includes(sub)
        idx = list(elem).index(child)
        elem.remove(child)
        for k, sc in enumerate(list(sub)):
          elem.insert(idx + k, sc)
      else:
        self._expand_includes(child)
      i += 1

  # -- defaults ---------------------------------------------------------------
  def _parse_defaults(self):
    self.classes = {}
    main = _DefaultClass('main', None)
    self.classes['main'] = main
    for d in self.root.findall('default'):
      self._parse_default_elem(d, main, top=True)

  def _parse_default_elem(self, elem, parent_cls, top=False):
    name = elem.attrib.get('class')
    if top and (name is None or name == 'main'):
      cls = parent_cls
    else:
      if name is None:
        raise MjcfError('nested <default> needs a class name')
      cls = _DefaultClass(name, parent_cls)
      self.classes[name] = cls
    # own element defaults first (so that children inherit them)
    for child in elem:
      if child.tag == 'default':
        continue
      tag = child.tag
      attrs = dict(child.attrib)
      if tag in _ACTUATOR_TAGS:
        attrs = _actuator_to_general(tag, attrs)
        tag = 'general'
      cls.attrs.setdefault(tag, {}).update(attrs)
    for child in elem:
      if child.tag == 'default':
        # child classes copy the parent's (now complete) defaults
        self._parse_default_elem(child, cls)

  def _resolve(self, tag, elem, childclass):
    cname = elem.attrib.get('class', childclass) or 'main'
    if cname not in self.classes:
      raise MjcfError('unknown default class %r' % cname)
    a = dict(self.classes[cname].get(tag))
    a.update(elem.attrib)
    return a

  # -- orientation ------------------------------------------------------------
  def _angle(self, x):
    return math.radians(x) if self.angle_deg else x

  def _orientation(self, a):
    if 'quat' in a:
      q = _vec(a['quat'], 4)
      n = np.linalg.norm(q)
      if n < MINVAL:
        raise MjcfError('zero quaternion')
      return q / n
    if 'axisangle' in a:
      v = _vec(a['axisangle'], 4)
      return axisangle_to_quat(v[:3], self._angle(v[3]))
    if 'euler' in a:
      e = _vec(a['euler'], 3)
      q = np.array([1.0, 0, 0, 0])
      for ch, ang in zip(self.eulerseq, e):
        ax = {'x': [1, 0, 0], 'y': [0, 1, 0], 'z': [0, 0, 1]}[ch.lower()]
        qi = axisangle_to_quat(ax, self._angle(ang))
        # lower-case: intrinsic (rotating frame) -> post-multiply
        q = quat_mul(q, qi) if ch.islower() else quat_mul(qi, q)
      return q / np.linalg.norm(q)
    if 'xyaxes' in a:
      v = _vec(a['xyaxes'], 6)
      x = v[:3] / np.linalg.norm(v[:3])
      y = v[3:] - x * np.dot(x, v[3:])
      y = y / np.linalg.norm(y)
      z = np.cross(x, y)
      return mat_to_quat(np.stack([x, y, z], axis=1))
    if 'zaxis' in a:
      return z_to_quat(_vec(a['zaxis'], 3))
    return np.array([1.0, 0, 0, 0])

  # -- top-level sections -----------------------------------------------------
  def _parse_compiler_option(self, model):
    for c in self.root.findall('compiler'):
      a = c.attrib
      if 'angle' in a:
        self.angle_deg = a['angle'] == 'degree'
      if 'eulerseq' in a:
        self.eulerseq = a['eulerseq']
      if 'settotalmass' in a:
        self.settotalmass = float(a['settotalmass'])
      if 'boundmass' in a:
        self.boundmass = float(a['boundmass'])
      if 'boundinertia' in a:
        self.boundinertia = float(a['boundinertia'])
      if 'autolimits' in a:
        self.autolimits = a['autolimits'] == 'true'
      if 'inertiafromgeom' in a:
        self.inertiafromgeom = a['inertiafromgeom']
      if a.get('coordinate', 'local') != 'local':
        raise MjcfError('only coordinate="local" is supported')
    o = model.opt
    for e in self.root.findall('option'):
      a = e.attrib
      for k in ('timestep', 'impratio', 'tolerance', 'ls_tolerance', 'noslip_tolerance'):
        if k in a:
          setattr(o, k, float(a[k]))
      for k in ('iterations', 'ls_iterations', 'noslip_iterations'):
        if k in a:
          setattr(o, k, int(a[k]))
      if 'gravity' in a:
        o.gravity = _vec(a['gravity'], 3)
      if 'integrator' in a:
        if a['integrator'] == 'implicit':
          # mjINT_IMPLICIT needs the Coriolis derivative (mjd_rne_vel) and an LU factorisation of a non-symmetric matrix:
          # refused by name rather than stepped with another integrator
          raise MjcfError('integrator="implicit" is not implemented (Euler, RK4 and implicitfast are)')
        if a['integrator'] not in ('Euler', 'RK4', 'implicitfast'):
          raise MjcfError('unknown integrator %r' % a['integrator'])
        o.integrator = {'Euler': 0, 'RK4': 1, 'implicitfast': 3}[a['integrator']]
      if 'cone' in a:
        o.cone = {'pyramidal': 0, 'elliptic': 1}[a['cone']]
      if 'solver' in a:
        o.solver = {'PGS': 0, 'CG': 1, 'Newton': 2}[a['solver']]
      for k in ('density', 'viscosity'):   # fluid forces: inertia-box model (no geom fluidshape)
        if k in a:
          setattr(o, k, float(a[k]))
          if getattr(o, k) < 0:
            raise MjcfError('option %s must be non-negative' % k)
      if 'wind' in a and np.any(_vec(a['wind'], 3) != 0):
        raise MjcfError('option wind is not supported')
      for f in e.findall('flag'):
        for k, v in f.attrib.items():
          if k in _DISABLE_FLAGS:
            bit = 1 << _DISABLE_FLAGS.index(k)
            if v == 'disable':
              o.disableflags |= bit
            else:
              o.disableflags &= ~bit
          elif k in _ENABLE_FLAGS:
            bit = 1 << _ENABLE_FLAGS.index(k)
            if v == 'enable':
              o.enableflags |= bit
            else:
              o.enableflags &= ~bit
          else:
            raise MjcfError('unknown flag %r' % k)

  # -- body tree ----------------------------------------------------------------
  def _parse_body(self, elem, parent_id, childclass):
    is_world = elem.tag == 'worldbody'
    if not is_world:
      childclass = elem.attrib.get('childclass', childclass)
    bid = len(self.bodies)
    body = dict(
        name=elem.attrib.get('name', 'world' if is_world else None),
        parent=parent_id,
        pos=_vec(elem.attrib['pos'], 3) if 'pos' in elem.attrib else np.zeros(3),
        quat=self._orientation(elem.attrib),
        joints=[], geoms=[], inertial=None, mocap=elem.attrib.get('mocap', 'false') == 'true')
    if body['mocap'] and (is_world or parent_id != 0):
      raise MjcfError('mocap body %r must be a child of the world body' % body['name'])
    if is_world:
      body['pos'] = np.zeros(3)
      body['quat'] = np.array([1.0, 0, 0, 0])
    self.bodies.append(body)
    for child in elem:
      tag = child.tag
      if tag == 'inertial':
        a = child.attrib
        inert = dict(pos=_vec(a['pos'], 3), quat=self._orientation(a), mass=float(a['mass']))
        if 'diaginertia' in a:
          inert['diag'] = _vec(a['diaginertia'], 3)
        elif 'fullinertia' in a:
          f = _vec(a['fullinertia'], 6)
          full = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
          w, v = np.linalg.eigh(full)
          order = np.argsort(-w)
          w, v = w[order], v[:, order]
          if np.linalg.det(v) < 0:
            v[:, 2] = -v[:, 2]
          inert['diag'] = w
          inert['quat'] = mat_to_quat(v)
        else:
          raise MjcfError('<inertial> needs diaginertia or fullinertia')
        body['inertial'] = inert
      elif tag in ('joint', 'freejoint'):
        if is_world:
          raise MjcfError('joints are not allowed in worldbody')
        if body['mocap']:
          raise MjcfError('mocap body %r cannot have joints' % body['name'])
        self._parse_joint(child, bid, childclass)
      elif tag == 'geom':
        self._parse_geom(child, bid, childclass)
      elif tag == 'site':
        self._parse_site(child, bid, childclass)
      elif tag == 'light':
        # rendering only: kept as host-side model arrays because tasks move lights (light_pos, suite/swimmer.py)
        cname = child.attrib.get('class', childclass) or 'main'
        if cname not in self.classes:
          raise MjcfError('unknown default class %r' % cname)
        a = dict(self.classes[cname].get('light') or {})
        a.update(child.attrib)
        self.lights.append(dict(name=a.get('name'), body=bid, pos=_vec(a['pos'], 3) if 'pos' in a else np.zeros(3),
                                dir=_vec(a['dir'], 3) if 'dir' in a else np.array([0.0, 0, -1])))
      elif tag == 'camera':
        # rendering only (MjModel.ncam, cam_* arrays): nothing on the device reads them
        cname = child.attrib.get('class', childclass) or 'main'
        if cname not in self.classes:
          raise MjcfError('unknown default class %r' % cname)
        a = dict(self.classes[cname].get('camera') or {})
        a.update(child.attrib)
        self.cameras.append(a.get('name'))
        self.camera_specs.append(dict(body=bid, pos=_vec(a['pos'], 3) if 'pos' in a else np.zeros(3),
                                      quat=self._orientation(a), fovy=float(a.get('fovy', 45)),
                                      mode=('fixed', 'track', 'trackcom', 'targetbody', 'targetbodycom').index(a.get('mode', 'fixed'))))
      elif tag == 'body':
        pass  # handled below so that this body's elements get ids first
      else:
        raise MjcfError('unsupported element <%s> in body' % tag)
    for child in elem:
      if child.tag == 'body':
        self._parse_body(child, bid, childclass)

  def _parse_joint(self, elem, bid, childclass):
    if elem.tag == 'freejoint':
      a = dict(elem.attrib)
      a['type'] = 'free'
    else:
      a = self._resolve('joint', elem, childclass)
    jtype = _JNT[a.get('type', 'hinge')]
    axis = _vec(a['axis'], 3) if 'axis' in a else np.array([0.0, 0, 1])
    n = np.linalg.norm(axis)
    if jtype in (_JNT['hinge'], _JNT['slide']):
      if n < MINVAL:
        raise MjcfError('zero joint axis')
      axis = axis / n
    else:
      axis = np.array([0.0, 0, 1])
    rng = _vec(a['range'], 2) if 'range' in a else np.zeros(2)
    limited = a.get('limited', 'auto')
    if limited == 'auto':
      lim = self.autolimits and 'range' in a
    else:
      lim = limited == 'true'
    if jtype == _JNT['free']:
      lim = False
    conv = self._angle if jtype in (_JNT['hinge'], _JNT['ball']) else (lambda x: x)
    rng = np.array([conv(rng[0]), conv(rng[1])])
    j = dict(
        name=a.get('name'), type=jtype, body=bid,
        pos=_vec(a['pos'], 3) if 'pos' in a else np.zeros(3),
        axis=axis, range=rng, limited=int(lim),
        ref=conv(float(a.get('ref', 0))) if jtype == _JNT['hinge'] else float(a.get('ref', 0)),
        springref=conv(float(a.get('springref', 0))) if jtype == _JNT['hinge'] else float(a.get('springref', 0)),
        stiffness=float(a.get('stiffness', 0)), damping=float(a.get('damping', 0)),
        armature=float(a.get('armature', 0)), frictionloss=float(a.get('frictionloss', 0)),
        margin=float(a.get('margin', 0)),
        solref=_vec(a.get('solreflimit', '0.02 1'), 2),
        solimp=_solimp(a.get('solimplimit', '0.9 0.95 0.001 0.5 2')),
        solreffriction=_vec(a.get('solreffriction', '0.02 1'), 2),
        solimpfriction=_solimp(a.get('solimpfriction', '0.9 0.95 0.001 0.5 2')))
    if jtype == _JNT['free']:
      j['pos'] = np.zeros(3)
    if lim and jtype in (_JNT['hinge'], _JNT['slide']) and rng[0] >= rng[1]:
      raise MjcfError('joint %r: range[0] must be < range[1]' % j['name'])
    self.bodies[bid]['joints'].append(len(self.joints))
    self.joints.append(j)

  def _parse_geom(self, elem, bid, childclass):
    a = self._resolve('geom', elem, childclass)
    gtype = _GEOM[a.get('type', 'sphere')]
    if gtype == _GEOM['hfield']:
      raise MjcfError('hfield geoms are not supported')
    mesh_verts = None
    if gtype == _GEOM['mesh']:
      # Scenery only: a mesh geom of the WORLD body that no moving geom can collide with would never reach the narrow
      # phase, so it compiles (frame at the centroid of its vertices, half-extents of its bounding box as `size`);
      # a mesh that could collide or that carries mass needs the convex-mesh narrow phase this backend does not have.
      if a.get('mesh') not in self.meshes:
        raise MjcfError('geom refers to unknown mesh %r' % a.get('mesh'))
      if bid != 0:
        raise MjcfError('mesh geoms are only supported as scenery of the world body (mesh collision is not implemented)')
      mesh_verts = self.meshes[a['mesh']]
    size = np.zeros(3)
    if 'size' in a:
      s = _vec(a['size'])
      size[:min(3, s.size)] = s[:3]
    pos = _vec(a['pos'], 3) if 'pos' in a else np.zeros(3)
    quat = self._orientation(a)
    if 'fromto' in a:
      if gtype not in (_GEOM['capsule'], _GEOM['cylinder'], _GEOM['box'], _GEOM['ellipsoid']):
        raise MjcfError('fromto requires capsule/cylinder/box/ellipsoid')
      ft = _vec(a['fromto'], 6)
      vec = ft[:3] - ft[3:]
      length = np.linalg.norm(vec)
      if length < MINVAL:
        raise MjcfError('fromto points too close')
      if gtype in (_GEOM['capsule'], _GEOM['cylinder']):
        size[1] = length / 2
      else:
        size[2] = length / 2
      pos = 0.5 * (ft[:3] + ft[3:])
      quat = z_to_quat(vec)
    if mesh_verts is not None:
      lo, hi = mesh_verts.min(axis=0), mesh_verts.max(axis=0)
      pos = pos + rot_vec(quat, 0.5 * (lo + hi))
      size = np.maximum(0.5 * (hi - lo), MINVAL)
    need = {_GEOM['sphere']: 1, _GEOM['capsule']: 2, _GEOM['cylinder']: 2,
            _GEOM['box']: 3, _GEOM['ellipsoid']: 3, _GEOM['plane']: 0, _GEOM['mesh']: 3}[gtype]
    if gtype != _GEOM['plane'] and np.any(size[:need] <= 0):
      raise MjcfError('geom %r: size must be positive' % a.get('name'))
    fr = np.array([1.0, 0.005, 0.0001])
    if 'friction' in a:
      f = _vec(a['friction'])
      fr[:min(3, f.size)] = f[:3]
    vol, inertia = _geom_volume_inertia(gtype, size)
    if 'mass' in a:
      mass = float(a['mass'])
      if vol > 0:
        inertia = inertia * (mass / vol)
      else:
        mass, inertia = 0.0, np.zeros(3)
    else:
      dens = float(a.get('density', 1000))
      mass, inertia = vol * dens, inertia * dens
    g = dict(
        name=a.get('name'), type=gtype, body=bid, size=size, pos=pos, quat=quat,
        contype=int(a.get('contype', 1)), conaffinity=int(a.get('conaffinity', 1)),
        condim=int(a.get('condim', 3)), priority=int(a.get('priority', 0)),
        group=int(a.get('group', 0)), friction=fr,
        solmix=float(a.get('solmix', 1)), solref=_vec(a.get('solref', '0.02 1'), 2),
        solimp=_solimp(a.get('solimp', '0.9 0.95 0.001 0.5 2')),
        margin=float(a.get('margin', 0)), gap=float(a.get('gap', 0)),
        mass=mass, inertia=inertia)
    g['rgba'] = _rgba(a)
    if a.get('material') in getattr(self, 'material_alpha', {}):
      g['invisible'] = int(self.material_alpha[a['material']] == 0)
    else:
      g['invisible'] = int(g['rgba'][3] == 0)
    if g['condim'] not in (1, 3, 4, 6):
      raise MjcfError('geom condim must be 1, 3, 4 or 6')
    self.bodies[bid]['geoms'].append(len(self.geoms))
    self.geoms.append(g)

  def _parse_site(self, elem, bid, childclass):
    a = self._resolve('site', elem, childclass)
    stype = _GEOM[a.get('type', 'sphere')]
    size = np.array([0.005, 0.005, 0.005])
    if 'size' in a:
      s = _vec(a['size'])
      size[:min(3, s.size)] = s[:3]
    pos = _vec(a['pos'], 3) if 'pos' in a else np.zeros(3)
    quat = self._orientation(a)
    if 'fromto' in a:
      ft = _vec(a['fromto'], 6)
      vec = ft[:3] - ft[3:]
      size[1] = np.linalg.norm(vec) / 2
      pos = 0.5 * (ft[:3] + ft[3:])
      quat = z_to_quat(vec)
    self.sites.append(dict(name=a.get('name'), type=stype, body=bid, size=size,
                           pos=pos, quat=quat, rgba=_rgba(a)))

  # -- actuators / sensors / contact / keyframes --------------------------------
  def _parse_actuators(self):
    for sec in self.root.findall('actuator'):
      for e in sec:
        if e.tag not in _ACTUATOR_TAGS:
          raise MjcfError('unsupported actuator <%s>' % e.tag)
        cname = e.attrib.get('class', 'main')
        if cname not in self.classes:
          raise MjcfError('unknown default class %r' % cname)
        a = dict(self.classes[cname].get('general'))
        a.update(_actuator_to_general(e.tag, dict(e.attrib)))
        if ('joint' in a) == ('tendon' in a):
          raise MjcfError('actuator %r: exactly one of joint= / tendon= transmissions is supported'
                          % a.get('name'))
        self.actuators.append(a)

  def _parse_tendons(self):
    """Fixed tendons (linear combinations of joint coordinates) used as actuator
    transmissions; anything that would add forces or constraints is rejected."""
    self.tendons = []
    for sec in self.root.findall('tendon'):
      for e in sec:
        if e.tag not in ('fixed', 'spatial'):
          raise MjcfError('unsupported tendon <%s>' % e.tag)
        cname = e.attrib.get('class', 'main')
        if cname not in self.classes:
          raise MjcfError('unknown default class %r' % cname)
        a = dict(self.classes[cname].get('tendon'))
        a.update(e.attrib)
        limited = a.get('limited', 'false') == 'true'
        if a.get('limited') == 'auto' or (self.autolimits and 'limited' not in a and 'range' in a):
          limited = 'range' in a
        if e.tag == 'spatial':
          # Spatial tendons: straight segments through sites.  A force-free one only draws a line
          # (suite/lqr.py:174-180) but keeps its row in the model (ntendon, names, data.ten_length); a length limit
          # (suite/ball_in_cup.xml) is supported.
          if not all(w.tag == 'site' for w in e) or len(e) < 2:
            raise MjcfError('spatial tendon %r: only site-to-site paths are supported' % a.get('name'))
          if any(float(a.get(k, 0)) != 0 for k in ('stiffness', 'damping', 'frictionloss')):
            raise MjcfError('spatial tendon %r: springs / dampers / friction are not supported' % a.get('name'))
          if any(act.get('tendon') == a.get('name') for act in self.actuators):
            raise MjcfError('spatial tendon %r: actuator transmissions are not supported' % a.get('name'))
          self.tendons.append(dict(name=a.get('name'), spatial=True, wraps=[(w.attrib['site'], 1.0) for w in e],
                                   stiffness=0.0, damping=0.0, limited=limited, attrs=a))
          continue
        if float(a.get('frictionloss', 0)) != 0:
          raise MjcfError('tendon %r: frictionloss is not supported' % a.get('name'))
        if 'springlength' in a:
          raise MjcfError('tendon %r: explicit springlength is not supported' % a.get('name'))
        wraps = []
        for w in e:
          if w.tag != 'joint':
            raise MjcfError('fixed tendon %r: unsupported element <%s>' % (a.get('name'), w.tag))
          wraps.append((w.attrib['joint'], float(w.attrib['coef'])))
        if not wraps:
          raise MjcfError('tendon %r is empty' % a.get('name'))
        self.tendons.append(dict(name=a.get('name'), spatial=False, wraps=wraps, stiffness=float(a.get('stiffness', 0)),
                                 damping=float(a.get('damping', 0)), limited=limited, attrs=a))
    # equality constraints: a single fixed tendon held at its reference length (plus polycoef[0])
    self.equalities = []
    for sec in self.root.findall('equality'):
      for e in sec:
        cname = e.attrib.get('class', 'main')
        if cname not in self.classes:
          raise MjcfError('unknown default class %r' % cname)
        a = dict(self.classes[cname].get('equality'))
        a.update(e.attrib)
        if e.tag not in ('tendon', 'connect', 'weld', 'joint') or 'tendon2' in a:
          raise MjcfError('unsupported equality constraint <%s> (connect, weld, joint and single-tendon equalities are supported)' % e.tag)
        a['_tag'] = e.tag
        self.equalities.append(a)

  def _parse_sensors(self):
    for sec in self.root.findall('sensor'):
      for e in sec:
        if e.tag not in _SENSORS:
          raise MjcfError('unsupported sensor <%s>' % e.tag)
        self.sensors.append((e.tag, dict(e.attrib)))

  def _parse_contact(self):
    for sec in self.root.findall('contact'):
      for e in sec:
        if e.tag == 'exclude':
          self.excludes.append((e.attrib['body1'], e.attrib['body2']))
        else:
          raise MjcfError('unsupported contact element <%s>' % e.tag)

  def _parse_keyframes(self):
    for sec in self.root.findall('keyframe'):
      for e in sec.findall('key'):
        self.keys.append(dict(e.attrib))

  # -- assemble -----------------------------------------------------------------
  def compile(self):
    m = Model()
    m.model_name = self.root.attrib.get('model', 'MuJoCo Model')
    self._parse_compiler_option(m)
    self._parse_defaults()
    wbs = self.root.findall('worldbody')
    if not wbs:      # MuJoCo compiles a model without one to the bare world body
      wbs = [ET.SubElement(self.root, 'worldbody')]
    # merge multiple worldbody sections
    wb = wbs[0]
    for extra in wbs[1:]:
      for c in list(extra):
        wb.append(c)
    # material alpha: rays (rangefinder sensors) skip invisible geoms (rgba alpha 0, or a material with alpha 0)
    self.material_alpha = {}
    self.materials = []
    for asset in self.root.findall('asset'):
      for mat in asset.findall('material'):
        rgba = _vec(mat.get('rgba', '1 1 1 1'))
        self.material_alpha[mat.get('name')] = float(rgba[3]) if rgba.size == 4 else 1.0
        self.materials.append((mat.get('name'), rgba if rgba.size == 4 else np.ones(4)))
      for mesh in asset.findall('mesh'):
        f = mesh.get('file')
        if f:
          name = mesh.get('name') or os.path.splitext(os.path.basename(f))[0]
          self.meshes[name] = _stl_vertices(self._asset_bytes(f)) * (_vec(mesh.get('scale', '1 1 1'), 3))
    self._parse_body(wb, -1, None)
    self._parse_actuators()
    self._parse_tendons()
    self._parse_sensors()
    self._parse_contact()
    self._parse_keyframes()
    self._assemble(m)
    return m

  def _assemble(self, m):
    nbody = len(self.bodies)
    njnt = len(self.joints)
    ngeom = len(self.geoms)
    nsite = len(self.sites)
    m.nbody, m.njnt, m.ngeom, m.nsite = nbody, njnt, ngeom, nsite
    # bodies are already in depth-first order with parent < child
    m.body_parentid = np.array([max(b['parent'], 0) for b in self.bodies])
    m.body_pos = np.array([b['pos'] for b in self.bodies]).reshape(nbody, 3)
    m.body_quat = np.array([b['quat'] for b in self.bodies]).reshape(nbody, 4)
    # joints -> qpos / dof addresses (joints are grouped by body in body order
    # because _parse_body visits a body's own elements before its children)
    order = []
    for b in self.bodies:
      order += b['joints']
    assert order == sorted(order)
    qposadr, dofadr = [], []
    nq = nv = 0
    for j in self.joints:
      qposadr.append(nq)
      dofadr.append(nv)
      nq += {0: 7, 1: 4, 2: 1, 3: 1}[j['type']]
      nv += {0: 6, 1: 3, 2: 1, 3: 1}[j['type']]
    m.nq, m.nv = nq, nv
    m.jnt_type = np.array([j['type'] for j in self.joints], dtype=np.int64)
    m.jnt_qposadr = np.array(qposadr, dtype=np.int64)
    m.jnt_dofadr = np.array(dofadr, dtype=np.int64)
    m.jnt_bodyid = np.array([j['body'] for j in self.joints], dtype=np.int64)
    m.jnt_limited = np.array([j['limited'] for j in self.joints], dtype=np.int64)
    m.jnt_pos = np.array([j['pos'] for j in self.joints]).reshape(njnt, 3)
    m.jnt_axis = np.array([j['axis'] for j in self.joints]).reshape(njnt, 3)
    m.jnt_stiffness = np.array([j['stiffness'] for j in self.joints], dtype=np.float64)
    m.jnt_range = np.array([j['range'] for j in self.joints]).reshape(njnt, 2)
    m.jnt_margin = np.array([j['margin'] for j in self.joints], dtype=np.float64)
    m.jnt_solref = np.array([j['solref'] for j in self.joints]).reshape(njnt, 2)
    m.jnt_solimp = np.array([j['solimp'] for j in self.joints]).reshape(njnt, 5)
    m.body_jntadr = np.full(nbody, -1, dtype=np.int64)
    m.body_jntnum = np.zeros(nbody, dtype=np.int64)
    m.body_dofadr = np.full(nbody, -1, dtype=np.int64)
    m.body_dofnum = np.zeros(nbody, dtype=np.int64)
    for bid, b in enumerate(self.bodies):
      if b['joints']:
        m.body_jntadr[bid] = b['joints'][0]
        m.body_jntnum[bid] = len(b['joints'])
        m.body_dofadr[bid] = dofadr[b['joints'][0]]
        m.body_dofnum[bid] = sum({0: 6, 1: 3, 2: 1, 3: 1}[self.joints[j]['type']]
                                 for j in b['joints'])
        types = [self.joints[j]['type'] for j in b['joints']]
        if _JNT['free'] in types and (len(types) > 1 or b['parent'] != 0):
          raise MjcfError('free joint must be alone in a top-level body')
    # dofs
    m.dof_bodyid = np.zeros(nv, dtype=np.int64)
    m.dof_jntid = np.zeros(nv, dtype=np.int64)
    m.dof_parentid = np.full(nv, -1, dtype=np.int64)
    m.dof_armature = np.zeros(nv)
    m.dof_damping = np.zeros(nv)
    m.dof_frictionloss = np.zeros(nv)
    m.dof_solref = np.zeros((nv, 2))
    m.dof_solimp = np.zeros((nv, 5))
    for jid, j in enumerate(self.joints):
      nd = {0: 6, 1: 3, 2: 1, 3: 1}[j['type']]
      for k in range(nd):
        d = dofadr[jid] + k
        m.dof_bodyid[d] = j['body']
        m.dof_jntid[d] = jid
        m.dof_armature[d] = j['armature']
        m.dof_damping[d] = j['damping']
        m.dof_frictionloss[d] = j['frictionloss']
        m.dof_solref[d] = j['solreffriction']
        m.dof_solimp[d] = j['solimpfriction']
    last_dof = np.full(nbody, -1, dtype=np.int64)  # last dof on the path to root
    for bid in range(1, nbody):
      prev = last_dof[m.body_parentid[bid]]
      if m.body_dofnum[bid]:
        for k in range(m.body_dofnum[bid]):
          d = m.body_dofadr[bid] + k
          m.dof_parentid[d] = prev
          prev = d
      last_dof[bid] = prev
    # weld / root ids
    m.body_weldid = np.zeros(nbody, dtype=np.int64)
    m.body_rootid = np.zeros(nbody, dtype=np.int64)
    for bid in range(1, nbody):
      p = m.body_parentid[bid]
      m.body_weldid[bid] = bid if m.body_dofnum[bid] else m.body_weldid[p]
      m.body_rootid[bid] = bid if p == 0 else m.body_rootid[p]
    # qpos0 / qpos_spring
    m.qpos0 = np.zeros(nq)
    m.qpos_spring = np.zeros(nq)
    for jid, j in enumerate(self.joints):
      a = qposadr[jid]
      if j['type'] == _JNT['free']:
        b = self.bodies[j['body']]
        m.qpos0[a:a+3] = b['pos']
        m.qpos0[a+3:a+7] = b['quat']
        m.qpos_spring[a:a+7] = m.qpos0[a:a+7]
      elif j['type'] == _JNT['ball']:
        m.qpos0[a:a+4] = [1, 0, 0, 0]
        m.qpos_spring[a:a+4] = [1, 0, 0, 0]
      else:
        m.qpos0[a] = j['ref']
        m.qpos_spring[a] = j['springref']
    # geoms
    m.geom_type = np.array([g['type'] for g in self.geoms], dtype=np.int64)
    m.geom_contype = np.array([g['contype'] for g in self.geoms], dtype=np.int64)
    m.geom_conaffinity = np.array([g['conaffinity'] for g in self.geoms], dtype=np.int64)
    m.geom_condim = np.array([g['condim'] for g in self.geoms], dtype=np.int64)
    m.geom_bodyid = np.array([g['body'] for g in self.geoms], dtype=np.int64)
    m.geom_priority = np.array([g['priority'] for g in self.geoms], dtype=np.int64)
    m.geom_invisible = np.array([g['invisible'] for g in self.geoms], dtype=np.int64)
    m.geom_size = np.array([g['size'] for g in self.geoms]).reshape(ngeom, 3)
    m.geom_pos = np.array([g['pos'] for g in self.geoms]).reshape(ngeom, 3)
    m.geom_quat = np.array([g['quat'] for g in self.geoms]).reshape(ngeom, 4)
    m.geom_friction = np.array([g['friction'] for g in self.geoms]).reshape(ngeom, 3)
    m.geom_solmix = np.array([g['solmix'] for g in self.geoms], dtype=np.float64)
    m.geom_solref = np.array([g['solref'] for g in self.geoms]).reshape(ngeom, 2)
    m.geom_solimp = np.array([g['solimp'] for g in self.geoms]).reshape(ngeom, 5)
    m.geom_margin = np.array([g['margin'] for g in self.geoms], dtype=np.float64)
    m.geom_gap = np.array([g['gap'] for g in self.geoms], dtype=np.float64)
    m.geom_rbound = np.array([_geom_rbound(g['type'], g['size']) for g in self.geoms],
                             dtype=np.float64)
    # geoms are appended body by body?  Not necessarily contiguous per body
    # (a body's geoms are parsed before its children), so they are: check.
    # mocap bodies: static children of the world whose pose is mjData.mocap_pos / mocap_quat (row body_mocapid)
    m.body_mocapid = np.full(nbody, -1, dtype=np.int64)
    for bid, b in enumerate(self.bodies):
      if b['mocap']:
        m.body_mocapid[bid] = int((m.body_mocapid >= 0).sum())
    m.nmocap = int((m.body_mocapid >= 0).sum())
    m.body_geomadr = np.full(nbody, -1, dtype=np.int64)
    m.body_geomnum = np.zeros(nbody, dtype=np.int64)
    for bid, b in enumerate(self.bodies):
      if b['geoms']:
        gs = b['geoms']
        assert gs == list(range(gs[0], gs[0] + len(gs)))
        m.body_geomadr[bid] = gs[0]
        m.body_geomnum[bid] = len(gs)
    # sites
    m.site_bodyid = np.array([s['body'] for s in self.sites], dtype=np.int64)
    m.site_type = np.array([s['type'] for s in self.sites], dtype=np.int64)
    m.site_size = np.array([s['size'] for s in self.sites]).reshape(nsite, 3)
    m.site_pos = np.array([s['pos'] for s in self.sites]).reshape(nsite, 3)
    # rendering attributes tasks write (suite/finger.py site_rgba, suite/fish.py geom_rgba, suite/swimmer.py light_pos,
    # suite/base.py mat_rgba): host-side arrays of the facade, never sent to the device
    m.site_rgba = np.array([s['rgba'] for s in self.sites], dtype=np.float64).reshape(nsite, 4)
    m.geom_rgba = np.array([g['rgba'] for g in self.geoms], dtype=np.float64).reshape(len(self.geoms), 4)
    m.nlight = len(self.lights)
    m.light_bodyid = np.array([l['body'] for l in self.lights], dtype=np.int64)
    m.light_pos = np.array([l['pos'] for l in self.lights], dtype=np.float64).reshape(m.nlight, 3)
    m.light_dir = np.array([l['dir'] for l in self.lights], dtype=np.float64).reshape(m.nlight, 3)
    m.nmat = len(self.materials)
    m.mat_rgba = np.array([r for _, r in self.materials], dtype=np.float64).reshape(m.nmat, 4)
    # mjModel's frame-coincidence flags: mj_kinematics shortcuts that a write to the frames must clear -- dm_control.mjcf
    # bindings zero them on every pos / quat write (mjcf/constants.py:52-61).  This backend composes every frame every
    # time, so they are plain host-side arrays that such writes may clear.
    m.body_sameframe = np.zeros(len(self.bodies), dtype=np.int64)
    m.body_simple = np.zeros(len(self.bodies), dtype=np.int64)
    m.geom_sameframe = np.zeros(len(self.geoms), dtype=np.int64)
    m.site_sameframe = np.zeros(nsite, dtype=np.int64)
    # sizes and names of what this backend does not simulate but MjModel counts (suite/suite_test.py:107-139 walks them)
    m.ncam = len(self.cameras)
    cs = self.camera_specs
    m.cam_bodyid = np.array([c['body'] for c in cs], dtype=np.int64)
    m.cam_mode = np.array([c['mode'] for c in cs], dtype=np.int64)
    m.cam_pos = np.array([c['pos'] for c in cs], dtype=np.float64).reshape(len(cs), 3)
    m.cam_quat = np.array([c['quat'] for c in cs], dtype=np.float64).reshape(len(cs), 4)
    m.cam_fovy = np.array([c['fovy'] for c in cs], dtype=np.float64)
    assets = [e for sec in self.root.findall('asset') for e in sec]
    custom = [e for sec in self.root.findall('custom') for e in sec]

    def named(elems, tag):
      out = []
      for e in elems:
        if e.tag == tag:
          f = e.get('file')
          out.append(e.get('name') or (os.path.splitext(os.path.basename(f))[0] if f else None))
      return out
    self._aux_names = dict(camera=list(self.cameras), mesh=named(assets, 'mesh'), hfield=named(assets, 'hfield'),
                           texture=named(assets, 'texture'), numeric=named(custom, 'numeric'), text=named(custom, 'text'),
                           tuple=named(custom, 'tuple'))
    m.nmesh, m.nhfield, m.ntex = (len(self._aux_names[k]) for k in ('mesh', 'hfield', 'texture'))
    m.nnumeric, m.ntext, m.ntuple = (len(self._aux_names[k]) for k in ('numeric', 'text', 'tuple'))
    # <custom><numeric>: host-side constants tasks read by name (mujoco/index.py:93-99 'nnumericdata'); `size` pads with zeros
    nums = []
    for e in custom:
      if e.tag == 'numeric':
        d = _vec(e.get('data', '0'))
        n = int(e.get('size', d.size))
        nums.append(np.concatenate([d, np.zeros(max(0, n - d.size))])[:max(n, 1)])
    m.numeric_size = np.array([a.size for a in nums], dtype=np.int64)
    m.numeric_adr = np.concatenate([[0], np.cumsum(m.numeric_size)])[:-1].astype(np.int64) if nums else np.zeros(0, dtype=np.int64)
    m.numeric_data = np.concatenate(nums) if nums else np.zeros(0)
    m.nnumericdata = int(m.numeric_data.size)
    m.mesh_vert = (np.concatenate([self.meshes[n] for n in self._aux_names['mesh'] if n in self.meshes])
                   if any(n in self.meshes for n in self._aux_names['mesh']) else np.zeros((0, 3)))
    m.site_quat = np.array([s['quat'] for s in self.sites]).reshape(nsite, 4)
    # inertial properties
    self._body_inertias(m)
    # names
    m.names = {
        'body': [b['name'] for b in self.bodies],
        'joint': [j['name'] for j in self.joints],
        'geom': [g['name'] for g in self.geoms],
        'site': [s['name'] for s in self.sites],
        'light': [l['name'] for l in self.lights],
        'material': [n for n, _ in self.materials],
    }
    m.names.update(self._aux_names)
    for kind, lst in m.names.items():
      named = [x for x in lst if x]
      if len(named) != len(set(named)):
        raise MjcfError('repeated %s name' % kind)
    self._actuators(m)
    self._sensors(m)
    self._pairs(m)
    self._keyframes(m)
    self._set_const(m)

  def _body_inertias(self, m):
    nbody = m.nbody
    m.body_mass = np.zeros(nbody)
    m.body_ipos = np.zeros((nbody, 3))
    m.body_iquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
    m.body_inertia = np.zeros((nbody, 3))
    for bid, b in enumerate(self.bodies):
      if bid == 0:
        continue
      use_geoms = (self.inertiafromgeom == 'true' or
                   (self.inertiafromgeom == 'auto' and b['inertial'] is None))
      if not use_geoms:
        if b['inertial'] is None:
          continue
        it = b['inertial']
        m.body_mass[bid] = it['mass']
        m.body_ipos[bid] = it['pos']
        m.body_iquat[bid] = it['quat']
        m.body_inertia[bid] = it['diag']
        continue
      gs = [self.geoms[g] for g in b['geoms']
            if self.inertiagrouprange[0] <= self.geoms[g]['group'] <= self.inertiagrouprange[1]]
      mass = sum(g['mass'] for g in gs)
      if mass <= 0:
        continue
      com = sum(g['mass'] * g['pos'] for g in gs) / mass
      full = np.zeros((3, 3))
      for g in gs:
        r = quat_to_mat(g['quat'])
        d = g['pos'] - com
        full += r @ np.diag(g['inertia']) @ r.T
        full += g['mass'] * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
      w, v = np.linalg.eigh(full)
      order = np.argsort(-w)
      w, v = w[order], v[:, order]
      if np.linalg.det(v) < 0:
        v[:, 2] = -v[:, 2]
      m.body_mass[bid] = mass
      m.body_ipos[bid] = com
      m.body_iquat[bid] = mat_to_quat(v)
      m.body_inertia[bid] = w
    if self.boundmass > 0:
      m.body_mass[1:] = np.maximum(m.body_mass[1:], self.boundmass)
    if self.boundinertia > 0:
      m.body_inertia[1:] = np.maximum(m.body_inertia[1:], self.boundinertia)
    if self.settotalmass > 0:
      tot = m.body_mass.sum()
      if tot > 0:
        s = self.settotalmass / tot
        m.body_mass *= s
        m.body_inertia *= s
    for bid in range(1, nbody):
      if m.body_dofnum[bid] and (m.body_mass[bid] < MINVAL or
                                 np.any(m.body_inertia[bid] < MINVAL)):
        # MuJoCo: "mass and inertia of moving bodies must be larger than mjMINVAL"
        # unless a descendant carries the mass; only reject the clear-cut case
        if not any(m.body_parentid[c] == bid for c in range(nbody)):
          raise MjcfError('moving body %r has no mass/inertia' % self.bodies[bid]['name'])
    m.body_subtreemass = m.body_mass.copy()
    for bid in range(nbody - 1, 0, -1):
      m.body_subtreemass[m.body_parentid[bid]] += m.body_subtreemass[bid]

  def _actuators(self, m):
    nu = len(self.actuators)
    m.nu, m.na = nu, 0
    m.actuator_trntype = np.zeros(nu, dtype=np.int64)
    m.actuator_dyntype = np.zeros(nu, dtype=np.int64)
    m.actuator_gaintype = np.zeros(nu, dtype=np.int64)
    m.actuator_biastype = np.zeros(nu, dtype=np.int64)
    m.actuator_trnid = np.full((nu, 2), -1, dtype=np.int64)
    m.actuator_ctrllimited = np.zeros(nu, dtype=np.int64)
    m.actuator_forcelimited = np.zeros(nu, dtype=np.int64)
    m.actuator_actlimited = np.zeros(nu, dtype=np.int64)
    m.actuator_actrange = np.zeros((nu, 2))
    m.actuator_gear = np.zeros((nu, 6))
    m.actuator_ctrlrange = np.zeros((nu, 2))
    m.actuator_forcerange = np.zeros((nu, 2))
    m.actuator_gainprm = np.zeros((nu, 10))
    m.actuator_biasprm = np.zeros((nu, 10))
    m.actuator_dynprm = np.zeros((nu, 10))
    # fixed tendons
    m.ntendon = len(self.tendons)
    m.tendon_adr = np.zeros(m.ntendon, dtype=np.int64)
    m.tendon_num = np.zeros(m.ntendon, dtype=np.int64)
    objid, prm, wtype = [], [], []
    for t, td in enumerate(self.tendons):
      m.tendon_adr[t] = len(objid)
      m.tendon_num[t] = len(td['wraps'])
      if td['spatial']:
        for sname, _ in td['wraps']:
          if sname not in m.names['site']:
            raise MjcfError('tendon %r refers to unknown site %r' % (td['name'], sname))
          objid.append(m.names['site'].index(sname))
          prm.append(0.0)
          wtype.append(C['DMC_WRAP_SITE'])
        continue
      for jname, coef in td['wraps']:
        wtype.append(C['DMC_WRAP_JOINT'])
        if jname not in m.names['joint']:
          raise MjcfError('tendon %r refers to unknown joint %r' % (td['name'], jname))
        jid = m.names['joint'].index(jname)
        if m.jnt_type[jid] not in (_JNT['hinge'], _JNT['slide']):
          raise MjcfError('fixed tendons may only wrap hinge/slide joints')
        objid.append(jid)
        prm.append(coef)
    m.nwrap = len(objid)
    m.wrap_objid = np.asarray(objid, dtype=np.int64)
    m.wrap_prm = np.asarray(prm, dtype=np.float64)
    m.wrap_type = np.asarray(wtype, dtype=np.int64)
    nt = m.ntendon
    m.tendon_limited = np.array([int(td['limited']) for td in self.tendons], dtype=np.int64)
    m.tendon_range = np.zeros((nt, 2))
    m.tendon_margin = np.zeros(nt)
    m.tendon_solref_lim = np.zeros((nt, 2))
    m.tendon_solimp_lim = np.zeros((nt, 5))
    m.tendon_invweight0 = np.zeros(nt)     # filled by _set_const
    for t, td in enumerate(self.tendons):
      a = td['attrs']
      if 'range' in a:
        m.tendon_range[t] = _vec(a['range'], 2)
      if td['limited'] and m.tendon_range[t, 0] >= m.tendon_range[t, 1]:
        raise MjcfError('tendon %r: range[0] must be < range[1]' % td['name'])
      m.tendon_margin[t] = float(a.get('margin', 0))
      m.tendon_solref_lim[t] = _vec(a.get('solreflimit', '0.02 1'), 2)
      m.tendon_solimp_lim[t] = _solimp(a.get('solimplimit', '0.9 0.95 0.001 0.5 2'))
    m.tendon_length0 = np.zeros(nt)          # filled by _set_const (length at qpos0)
    m.neq = len(self.equalities)
    m.eq_type = np.full(m.neq, C['DMC_EQ_TENDON'], dtype=np.int64)
    m.eq_obj1id = np.zeros(m.neq, dtype=np.int64)
    m.eq_obj2id = np.full(m.neq, -1, dtype=np.int64)
    m.eq_active0 = np.ones(m.neq, dtype=np.int64)
    m.eq_solref = np.zeros((m.neq, 2))
    m.eq_solimp = np.zeros((m.neq, 5))
    m.eq_data = np.zeros((m.neq, 11))
    eq_names = []
    self._eq_auto = {}      # weld / connect: which parts of eq_data _set_const derives from the pose at qpos0
    for k, a in enumerate(self.equalities):
      eq_names.append(a.get('name'))
      tag = a['_tag']
      m.eq_active0[k] = int(a.get('active', 'true') == 'true')
      m.eq_solref[k] = _vec(a.get('solref', '0.02 1'), 2)
      m.eq_solimp[k] = _solimp(a.get('solimp', '0.9 0.95 0.001 0.5 2'))
      if tag == 'tendon':
        tname = a.get('tendon1')
        names_t = [td['name'] for td in self.tendons]
        if tname not in names_t:
          raise MjcfError('equality refers to unknown tendon %r' % tname)
        t = names_t.index(tname)
        if self.tendons[t]['spatial']:
          raise MjcfError('equality on a spatial tendon is not supported')
        m.eq_obj1id[k] = t
        pc = _vec(a.get('polycoef', '0 1 0 0 0'))
        m.eq_data[k, :pc.size] = pc
      elif tag == 'joint':
        m.eq_type[k] = C['DMC_EQ_JOINT']
        for key, dst in (('joint1', m.eq_obj1id), ('joint2', m.eq_obj2id)):
          if key in a:
            if a[key] not in m.names['joint']:
              raise MjcfError('equality refers to unknown joint %r' % a[key])
            j = m.names['joint'].index(a[key])
            if m.jnt_type[j] not in (_JNT['hinge'], _JNT['slide']):
              raise MjcfError('joint equalities need hinge / slide joints')
            dst[k] = j
          elif key == 'joint1':
            raise MjcfError('joint equality needs joint1')
        pc = _vec(a.get('polycoef', '0 1 0 0 0'))
        m.eq_data[k, :pc.size] = pc
      else:
        m.eq_type[k] = C['DMC_EQ_CONNECT'] if tag == 'connect' else C['DMC_EQ_WELD']
        if 'body1' not in a:
          raise MjcfError('%s equality needs body1 (site-based definitions are not supported)' % tag)
        for key, dst in (('body1', m.eq_obj1id), ('body2', m.eq_obj2id)):
          name = a.get(key)
          if name is None:
            dst[k] = 0                         # the world
          elif name not in m.names['body']:
            raise MjcfError('equality refers to unknown body %r' % name)
          else:
            dst[k] = m.names['body'].index(name)
        anchor = _vec(a['anchor'], 3) if 'anchor' in a else np.zeros(3)
        if tag == 'connect':
          if 'anchor' not in a:
            raise MjcfError('connect equality needs an anchor')
          m.eq_data[k, 0:3] = anchor           # in the body1 frame; the body2 one follows from qpos0
          self._eq_auto[k] = 'connect'
        else:
          m.eq_data[k, 0:3] = anchor           # in the body2 frame
          m.eq_data[k, 10] = float(a.get('torquescale', 1))
          rel = _vec(a.get('relpose', '0 1 0 0 0 0 0'), 7)
          if np.any(rel[3:] != 0):
            q = rel[3:] / np.linalg.norm(rel[3:])
            m.eq_data[k, 6:10] = q
            # anchor expressed in body1: relpose maps body2 coordinates into body1's frame
            m.eq_data[k, 3:6] = rel[:3] + quat_to_mat(q) @ anchor
            self._eq_auto[k] = None
          else:
            self._eq_auto[k] = 'weld'          # pose of body2 in body1 taken from qpos0
    m.names['equality'] = eq_names
    m.tendon_stiffness = np.array([td['stiffness'] for td in self.tendons], dtype=np.float64)
    m.tendon_damping = np.array([td['damping'] for td in self.tendons], dtype=np.float64)
    # spring rest length: the tendon's length at qpos0 (springlength = -1 default)
    m.tendon_lengthspring = np.array([0.0 if td['spatial'] else
                                      sum(c * m.qpos0[m.jnt_qposadr[m.names['joint'].index(j)]] for j, c in td['wraps'])
                                      for td in self.tendons], dtype=np.float64)
    m.names['tendon'] = [td['name'] for td in self.tendons]
    names = []
    for i, a in enumerate(self.actuators):
      names.append(a.get('name'))
      if 'tendon' in a:
        if a['tendon'] not in m.names['tendon']:
          raise MjcfError('actuator refers to unknown tendon %r' % a['tendon'])
        m.actuator_trntype[i] = C['DMC_TRN_TENDON']
        m.actuator_trnid[i, 0] = m.names['tendon'].index(a['tendon'])
      else:
        jname = a['joint']
        if jname not in m.names['joint']:
          raise MjcfError('actuator refers to unknown joint %r' % jname)
        jid = m.names['joint'].index(jname)
        if m.jnt_type[jid] not in (_JNT['hinge'], _JNT['slide']):
          raise MjcfError('actuators on ball/free joints are not supported')
        m.actuator_trnid[i, 0] = jid
      dyn = a.get('dyntype', 'none')
      if dyn not in ('none', 'integrator', 'filter', 'filterexact'):
        raise MjcfError('actuator dyntype %r is not supported' % dyn)
      m.actuator_dyntype[i] = {'none': 0, 'integrator': 1, 'filter': 2, 'filterexact': 3}[dyn]
      if a.get('actearly', 'false') == 'true':
        raise MjcfError('actuator %r: actearly is not supported' % a.get('name'))
      m.actuator_gaintype[i] = {'fixed': 0, 'affine': 1}[a.get('gaintype', 'fixed')]
      m.actuator_biastype[i] = {'none': 0, 'affine': 1}[a.get('biastype', 'none')]
      g = _vec(a.get('gear', '1'))
      m.actuator_gear[i, :g.size] = g
      gp = _vec(a.get('gainprm', '1'))
      m.actuator_gainprm[i, :gp.size] = gp
      bp = _vec(a.get('biasprm', '0'))
      m.actuator_biasprm[i, :bp.size] = bp
      m.actuator_dynprm[i, 0] = 1
      if 'dynprm' in a:
        dp = _vec(a['dynprm'])
        m.actuator_dynprm[i, :dp.size] = dp
      for key, lim, rng in (('ctrl', m.actuator_ctrllimited, m.actuator_ctrlrange),
                            ('force', m.actuator_forcelimited, m.actuator_forcerange),
                            ('act', m.actuator_actlimited, m.actuator_actrange)):
        has = (key + 'range') in a
        if has:
          rng[i] = _vec(a[key + 'range'], 2)
        flag = a.get(key + 'limited', 'auto')
        lim[i] = int((self.autolimits and has) if flag == 'auto' else flag == 'true')
      if m.actuator_actlimited[i]:
        if m.actuator_dyntype[i] == 0:
          raise MjcfError('actuator %r: actlimited needs a dyntype (the actuator has no activation state)' % a.get('name'))
        if not m.actuator_actrange[i, 0] < m.actuator_actrange[i, 1]:
          raise MjcfError('actuator %r: actrange[0] must be < actrange[1]' % a.get('name'))
    m.names['actuator'] = names
    # one activation state per actuator with dynamics, in actuator order (mjModel.actuator_actadr)
    m.na = int(np.count_nonzero(m.actuator_dyntype))
    if m.na and m.opt.integrator not in (0, 3):
      raise MjcfError('actuator dynamics are only supported with the Euler and implicitfast integrators')
    if m.opt.integrator == 3:
      # implicitfast: the velocity derivatives this backend folds into the integration matrix are the DIAGONAL ones
      # (joint damping, joint-transmission actuators).  Terms that couple dofs -- damped tendons, tendon-transmission
      # actuators with a velocity-dependent bias / gain -- and the fluid-force derivatives are refused, not dropped.
      if m.opt.density > 0 or m.opt.viscosity > 0:
        raise MjcfError('integrator="implicitfast" with fluid forces (density / viscosity) is not implemented')
      if m.ntendon and np.any(np.asarray(m.tendon_damping) > 0):
        raise MjcfError('integrator="implicitfast" with damped tendons is not implemented')
      for i in range(nu):
        vel_term = (m.actuator_biastype[i] == 1 and m.actuator_biasprm[i, 2] != 0) or \
                   (m.actuator_gaintype[i] == 1 and m.actuator_gainprm[i, 2] != 0)
        if vel_term and m.actuator_trntype[i] != 0:
          raise MjcfError('integrator="implicitfast": actuator %r has a velocity-dependent force on a tendon '
                          'transmission, which is not implemented' % names[i])

  def _sensors(self, m):
    ns = len(self.sensors)
    m.nsensor = ns
    m.sensor_type = np.zeros(ns, dtype=np.int64)
    m.sensor_objtype = np.zeros(ns, dtype=np.int64)
    m.sensor_objid = np.zeros(ns, dtype=np.int64)
    m.sensor_adr = np.zeros(ns, dtype=np.int64)
    m.sensor_dim = np.zeros(ns, dtype=np.int64)
    m.sensor_needstage = np.zeros(ns, dtype=np.int64)
    m.sensor_reftype = np.zeros(ns, dtype=np.int64)
    m.sensor_refid = np.full(ns, -1, dtype=np.int64)
    m.sensor_cutoff = np.zeros(ns)
    adr = 0
    names = []
    kind = {C['DMC_OBJ_SITE']: 'site', C['DMC_OBJ_BODY']: 'body', C['DMC_OBJ_XBODY']: 'body',
            C['DMC_OBJ_GEOM']: 'geom', C['DMC_OBJ_JOINT']: 'joint', C['DMC_OBJ_ACTUATOR']: 'actuator'}
    for i, (tag, a) in enumerate(self.sensors):
      stype, attr, objtype, dim, stage = _SENSORS[tag]
      reftype = refname = None
      if objtype is None:
        frames = {'body': C['DMC_OBJ_BODY'], 'xbody': C['DMC_OBJ_XBODY'], 'geom': C['DMC_OBJ_GEOM'], 'site': C['DMC_OBJ_SITE']}
        try:
          objtype = frames[a.get('objtype')]
        except KeyError:
          raise MjcfError('sensor %r: unsupported objtype %r' % (a.get('name'), a.get('objtype')))
        if 'reftype' in a or 'refname' in a:
          # the object's frame expressed in a reference frame (mj_sensorPos: R_ref' (p - p_ref), R_ref' axis,
          # conj(q_ref) q; mj_sensorVel: R_ref' (v - v_ref + r x w_ref), R_ref' (w - w_ref))
          try:
            reftype = frames[a.get('reftype')]
          except KeyError:
            raise MjcfError('sensor %r: unsupported reftype %r' % (a.get('name'), a.get('reftype')))
          refname = a.get('refname')
      names.append(a.get('name'))
      oname = a.get(attr)
      lst = m.names[kind[objtype]]
      if oname not in lst:
        raise MjcfError('sensor %r refers to unknown %s %r' % (a.get('name'), attr, oname))
      m.sensor_type[i] = stype
      m.sensor_objtype[i] = objtype
      m.sensor_objid[i] = lst.index(oname)
      m.sensor_adr[i] = adr
      m.sensor_dim[i] = dim
      m.sensor_needstage[i] = stage
      if reftype is not None:
        rl = m.names[kind[reftype]]
        if refname not in rl:
          raise MjcfError('sensor %r refers to unknown reference %r' % (a.get('name'), refname))
        m.sensor_reftype[i] = reftype
        m.sensor_refid[i] = rl.index(refname)
      m.sensor_cutoff[i] = float(a.get('cutoff', 0))
      adr += dim
    m.nsensordata = adr
    m.names['sensor'] = names

  def _pairs(self, m):
    m.exclude_bodies = []
    for b1, b2 in self.excludes:
      i1, i2 = m.names['body'].index(b1), m.names['body'].index(b2)
      m.exclude_bodies.append((min(i1, i2), max(i1, i2)))
    candidate_pairs(m)

  def _keyframes(self, m):
    nkey = len(self.keys)
    m.nkey = nkey
    m.key_qpos = np.tile(m.qpos0, (nkey, 1)).reshape(nkey, m.nq)
    m.key_qvel = np.zeros((nkey, m.nv))
    m.key_ctrl = np.zeros((nkey, m.nu))
    # the rest of what mj_resetDataKeyframe restores: time, activations, mocap poses (default: the model's body poses)
    m.key_time = np.zeros(nkey)
    m.key_act = np.zeros((nkey, m.na))
    mocap_bodies = [b for b in range(m.nbody) if m.body_mocapid[b] >= 0]
    mocap_bodies.sort(key=lambda b: m.body_mocapid[b])
    m.key_mpos = np.tile(m.body_pos[mocap_bodies].reshape(-1), (nkey, 1)).reshape(nkey, 3 * m.nmocap)
    m.key_mquat = np.tile(m.body_quat[mocap_bodies].reshape(-1), (nkey, 1)).reshape(nkey, 4 * m.nmocap)
    names = []
    for i, k in enumerate(self.keys):
      names.append(k.get('name'))
      if 'qpos' in k:
        m.key_qpos[i] = _vec(k['qpos'], m.nq)
      if 'qvel' in k:
        m.key_qvel[i] = _vec(k['qvel'], m.nv)
      if 'ctrl' in k:
        m.key_ctrl[i] = _vec(k['ctrl'], m.nu)
      if 'time' in k:
        m.key_time[i] = float(k['time'])
      if 'act' in k:
        m.key_act[i] = _vec(k['act'], m.na)
      if 'mpos' in k:
        m.key_mpos[i] = _vec(k['mpos'], 3 * m.nmocap)
      if 'mquat' in k:
        q = _vec(k['mquat'], 4 * m.nmocap).reshape(m.nmocap, 4)
        m.key_mquat[i] = (q / np.linalg.norm(q, axis=1, keepdims=True)).reshape(-1)
    m.names['key'] = names

  # -- constants evaluated at qpos0 -------------------------------------------
  def _set_const(self, m):
    """body_invweight0 / dof_invweight0 / stat.meaninertia: inverse inertia seen
    at qpos0, from a dense M built with world-frame
    body Jacobians."""
    nbody, nv = m.nbody, m.nv
    m.body_invweight0 = np.zeros((nbody, 2))
    m.dof_invweight0 = np.zeros(nv)
    if nv == 0:
      m.stat_meaninertia = 1.0
      return
    xpos = np.zeros((nbody, 3))
    xquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
    dof_axis = np.zeros((nv, 3))     # rotation axis (world) or translation dir
    dof_anchor = np.zeros((nv, 3))
    dof_rot = np.zeros(nv, dtype=bool)
    for b in range(1, nbody):
      p = m.body_parentid[b]
      jn, ja = m.body_jntnum[b], m.body_jntadr[b]
      if jn == 1 and m.jnt_type[ja] == _JNT['free']:
        a = m.jnt_qposadr[ja]
        xpos[b] = m.qpos0[a:a+3]
        q = m.qpos0[a+3:a+7]
        xquat[b] = q / np.linalg.norm(q)
      else:
        xpos[b] = xpos[p] + rot_vec(xquat[p], m.body_pos[b])
        xquat[b] = quat_mul(xquat[p], m.body_quat[b])
        # at qpos0 hinge/slide displacements are zero, ball quats identity
      mat = quat_to_mat(xquat[b])
      for j in range(ja, ja + jn):
        d = m.jnt_dofadr[j]
        t = m.jnt_type[j]
        anchor = xpos[b] + mat @ m.jnt_pos[j]
        if t == _JNT['free']:
          for k in range(3):
            dof_axis[d+k] = np.eye(3)[k]
            dof_rot[d+k] = False
            dof_axis[d+3+k] = mat[:, k]
            dof_anchor[d+3+k] = xpos[b]
            dof_rot[d+3+k] = True
        elif t == _JNT['ball']:
          for k in range(3):
            dof_axis[d+k] = mat[:, k]
            dof_anchor[d+k] = anchor
            dof_rot[d+k] = True
        elif t == _JNT['hinge']:
          dof_axis[d] = mat @ m.jnt_axis[j]
          dof_anchor[d] = anchor
          dof_rot[d] = True
        else:
          dof_axis[d] = mat @ m.jnt_axis[j]
          dof_rot[d] = False
    xipos = np.zeros((nbody, 3))
    jacp = np.zeros((nbody, 3, nv))
    jacr = np.zeros((nbody, 3, nv))
    mass_matrix = np.diag(m.dof_armature.astype(np.float64))
    for b in range(1, nbody):
      mat = quat_to_mat(xquat[b])
      xipos[b] = xpos[b] + mat @ m.body_ipos[b]
      imat = quat_to_mat(quat_mul(xquat[b], m.body_iquat[b]))
      # dofs on the path to the root
      d = m.body_dofadr[b] + m.body_dofnum[b] - 1 if m.body_dofnum[b] else -1
      if d < 0:
        pb = m.body_parentid[b]
        while pb > 0 and not m.body_dofnum[pb]:
          pb = m.body_parentid[pb]
        d = m.body_dofadr[pb] + m.body_dofnum[pb] - 1 if pb > 0 else -1
      while d >= 0:
        if dof_rot[d]:
          jacr[b, :, d] = dof_axis[d]
          jacp[b, :, d] = np.cross(dof_axis[d], xipos[b] - dof_anchor[d])
        else:
          jacp[b, :, d] = dof_axis[d]
        d = m.dof_parentid[d]
      iw = imat @ np.diag(m.body_inertia[b]) @ imat.T
      mass_matrix += m.body_mass[b] * jacp[b].T @ jacp[b] + jacr[b].T @ iw @ jacr[b]
    self.mass_matrix0 = mass_matrix
    minv = _robust_inverse(mass_matrix)
    diag = np.diag(minv).copy()
    for j in range(m.njnt):
      d = m.jnt_dofadr[j]
      if m.jnt_type[j] == _JNT['free']:
        diag[d:d+3] = diag[d:d+3].mean()
        diag[d+3:d+6] = diag[d+3:d+6].mean()
      elif m.jnt_type[j] == _JNT['ball']:
        diag[d:d+3] = diag[d:d+3].mean()
    m.dof_invweight0 = diag
    for b in range(1, nbody):
      if m.body_weldid[b] == 0:
        continue
      ap = jacp[b] @ minv @ jacp[b].T
      ar = jacr[b] @ minv @ jacr[b].T
      m.body_invweight0[b, 0] = max(MINVAL, np.trace(ap) / 3)
      m.body_invweight0[b, 1] = max(MINVAL, np.trace(ar) / 3)
    m.stat_meaninertia = float(np.mean(np.diag(mass_matrix)))
    # tendon_invweight0 = J M^-1 J' of the tendon length Jacobian at qpos0
    for t in range(getattr(m, 'ntendon', 0)):
      J = np.zeros(nv)
      w0, wn = m.tendon_adr[t], m.tendon_num[t]
      if wn and m.wrap_type[w0] == C['DMC_WRAP_JOINT']:
        for w in range(w0, w0 + wn):
          J[m.jnt_dofadr[m.wrap_objid[w]]] += m.wrap_prm[w]
      else:
        def point_jac(sid):
          b = m.site_bodyid[sid]
          p = xpos[b] + quat_to_mat(xquat[b]) @ m.site_pos[sid]
          jp = np.zeros((3, nv))
          d = -1
          bb = b
          while bb > 0 and d < 0:
            if m.body_dofnum[bb]:
              d = m.body_dofadr[bb] + m.body_dofnum[bb] - 1
            bb = m.body_parentid[bb]
          while d >= 0:
            jp[:, d] = np.cross(dof_axis[d], p - dof_anchor[d]) if dof_rot[d] else dof_axis[d]
            d = m.dof_parentid[d]
          return p, jp
        for w in range(w0, w0 + wn - 1):
          p0, j0 = point_jac(m.wrap_objid[w])
          p1, j1 = point_jac(m.wrap_objid[w + 1])
          dvec = p1 - p0
          n = np.linalg.norm(dvec)
          if n > MINVAL:
            J += (dvec / n) @ (j1 - j0)
      m.tendon_invweight0[t] = max(MINVAL, float(J @ minv @ J))
      if wn and m.wrap_type[w0] == C['DMC_WRAP_JOINT']:
        m.tendon_length0[t] = sum(m.wrap_prm[w] * m.qpos0[m.jnt_qposadr[m.wrap_objid[w]]] for w in range(w0, w0 + wn))
    # connect / weld equalities: the parts of eq_data that follow from the reference pose (qpos0)
    for k, kind in getattr(self, '_eq_auto', {}).items():
      b1, b2 = int(m.eq_obj1id[k]), int(m.eq_obj2id[k])
      R1, R2 = quat_to_mat(xquat[b1]), quat_to_mat(xquat[b2])
      if kind == 'connect':
        world = xpos[b1] + R1 @ m.eq_data[k, 0:3]
        m.eq_data[k, 3:6] = R2.T @ (world - xpos[b2])
      elif kind == 'weld':
        world = xpos[b2] + R2 @ m.eq_data[k, 0:3]
        m.eq_data[k, 3:6] = R1.T @ (world - xpos[b1])
        q1 = xquat[b1]
        m.eq_data[k, 6:10] = quat_mul(np.array([q1[0], -q1[1], -q1[2], -q1[3]]), xquat[b2])


def _solimp(s):
  v = _vec(s)
  out = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
  out[:min(5, v.size)] = v[:5]
  return out


def _robust_inverse(mat):
  """Inverse of the mass matrix at qpos0.  A model may legitimately have a singular one (three hinges with the
  same axis on one body and no armature: suite/utils/randomizers_test.py); MuJoCo's factorisation clamps the
  pivots at mjMINVAL instead of failing, so does this fallback (Cholesky with clamped pivots)."""
  try:
    return np.linalg.inv(mat)
  except np.linalg.LinAlgError:
    n = mat.shape[0]
    low = np.zeros_like(mat)
    for j in range(n):
      d = mat[j, j] - low[j, :j] @ low[j, :j]
      low[j, j] = np.sqrt(max(d, MINVAL))
      for i in range(j + 1, n):
        low[i, j] = (mat[i, j] - low[i, :j] @ low[j, :j]) / low[j, j]
    linv = np.linalg.inv(low)
    return linv.T @ linv


_COMPILE_CACHE = collections.OrderedDict()
_COMPILE_LOCK = threading.Lock()
_COMPILE_CACHE_SIZE = 16


def candidate_pairs(m):
  """Static candidate geom pairs = MuJoCo's body-pair/geom-pair filters that
  do not depend on the state: same/welded body,
  parent-child (filterparent), contype/conaffinity, <exclude>.  Writes m.npair / pair_geom1 / pair_geom2 (also called
  by the Physics facade when a task rewrites geom_contype / geom_conaffinity at run time)."""
  excl = set(getattr(m, 'exclude_bodies', ()))
  filterparent = not (m.opt.disableflags & C['DMC_DSBL_FILTERPARENT'])
  weld = m.body_weldid
  pairs = []
  for b1 in range(m.nbody):
    for b2 in range(b1 + 1, m.nbody):
      if not m.body_geomnum[b1] or not m.body_geomnum[b2]:
        continue
      if (b1, b2) in excl:
        continue
      w1, w2 = weld[b1], weld[b2]
      if w1 == w2:
        continue
      wp1 = weld[m.body_parentid[w1]]
      wp2 = weld[m.body_parentid[w2]]
      if filterparent and w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
        continue
      for g1 in range(m.body_geomadr[b1], m.body_geomadr[b1] + m.body_geomnum[b1]):
        for g2 in range(m.body_geomadr[b2], m.body_geomadr[b2] + m.body_geomnum[b2]):
          if not ((m.geom_contype[g1] & m.geom_conaffinity[g2]) or
                  (m.geom_contype[g2] & m.geom_conaffinity[g1])):
            continue
          t1, t2 = m.geom_type[g1], m.geom_type[g2]
          if t1 == _GEOM['plane'] and t2 == _GEOM['plane']:
            continue
          if _GEOM['mesh'] in (t1, t2):
            raise MjcfError('a mesh geom is in contact range of a moving geom: mesh collision is not implemented')
          pairs.append((g2, g1) if t1 > t2 else (g1, g2))
  m.npair = len(pairs)
  m.pair_geom1 = np.array([p[0] for p in pairs], dtype=np.int64)
  m.pair_geom2 = np.array([p[1] for p in pairs], dtype=np.int64)


def compile_xml(xml_string, assets=None, cache=True):
  """MJCF string (+ include assets) -> Model.  Mirrors
  `MjModel.from_xml_string(xml_string, assets)` (wrapper/core.py:289).

  Compiled models are cached by the hash of the XML and its assets: composer environments
  recompile their model at every episode (composer/environment.py:377-383), mostly to the very
  same XML.  A cache hit returns a private deep copy (tasks write into model fields)."""
  if isinstance(xml_string, bytes):
    xml_string = xml_string.decode()
  if not cache:
    return _Compiler(xml_string, assets).compile()
  h = hashlib.sha1(xml_string.encode())
  for k in sorted(assets or {}):
    v = assets[k]
    h.update(b'\0' + k.encode() + b'\0' + (v if isinstance(v, bytes) else v.encode()))
  key = h.digest()
  with _COMPILE_LOCK:      # (models are loaded from several threads at once: mujoco/thread_safety_test.py:53-75)
    hit = _COMPILE_CACHE.get(key)
    if hit is not None:
      _COMPILE_CACHE.move_to_end(key)
  if hit is None:
    hit = _Compiler(xml_string, assets).compile()
    with _COMPILE_LOCK:
      _COMPILE_CACHE[key] = hit
      while len(_COMPILE_CACHE) > _COMPILE_CACHE_SIZE:
        _COMPILE_CACHE.popitem(last=False)
  return copy.deepcopy(hit)
