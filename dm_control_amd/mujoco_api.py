"""The `(MjModel, MjData)` seam: the part of the `mujoco` Python package that dm_control calls, on the HIP batch.

The reference crosses into native code through pybind11 functions that take `(model.ptr, data.ptr)`:

  mujoco.mj_step / mj_step1 / mj_step2          dm_control/mujoco/engine.py:156-162,176
  mujoco.mj_forward                             engine.py:343
  mujoco.mj_resetData / mj_resetDataKeyframe    engine.py:318,323
  mujoco.mj_stateSize / mj_getState / mj_setState   engine.py:248-249,280
  mujoco.MjModel.from_xml_string / from_xml_path / from_binary_path, mj_saveModel, mj_sizeModel, mj_saveLastXML
                                                dm_control/mujoco/wrapper/core.py:180-182,205,233-234,325-331
  mujoco.MjData(model), mj_name2id / mj_id2name, mju_str2Type / mju_type2Str   core.py:85,92,364,387,475
  mujoco.mj_objectVelocity, mj_fwdActuation / mj_fwdAcceleration / mj_fwdConstraint, mj_contactForce   core.py:522,546-551
  mujoco.mj_subtreeVel                          dm_control/locomotion/walkers/legacy_base.py:148,186
  mujoco.mju_sym2dense                          dm_control/suite/lqr_solver.py:49-51
  mujoco.set_mjcb_* / get_mjcb_*, set_mju_user_warning   core.py:74,98-141

This module offers those names with those signatures.  `MjModel` wraps a model compiled by `mjcf_compiler`; every
`MjData` owns ONE environment of a `dmc_batch` (include/dmc_batch.h, batch size 1, fp64) on the GPU.  With it installed as
`mujoco` (`install()`), the reference's own `dm_control/mujoco/engine.py`, `wrapper/core.py` and `index.py` run UNMODIFIED
on `libdmc_hip.so` (tests/test_reference_mujoco.py runs their unit tests that way).

mjData memory.  MuJoCo's arrays are views into one C struct that the engine rewrites in place, and the reference relies on
that (`engine.Physics` keeps `data.warning.number` and compares it after a step; `index.FieldIndexer` keeps a weak proxy of
every array).  Here every array handed out is ONE host ndarray for the life of the MjData:

  * input fields (`qpos qvel act ctrl qacc_warmstart qfrc_applied xfrc_applied mocap_pos mocap_quat time eq_active`) are
    compared with what the device holds before every launch and uploaded where they differ;
  * after every launch, every array that has been handed out is rewritten in place from the device -- device outputs by
    one batched read, and the mjData arrays the fused kernel keeps on chip and never stores (`ximat xanchor xaxis M qM qLD
    qLDiagInv energy subtree_linvel subtree_angmom cam_x* light_x* ten_length ten_velocity wrap_xpos act_dot`) derived on the
    host from the device's kinematics;
  * model arrays are the compiled model's own ndarrays, all writable: before a launch the model blob is re-packed and
    compared with what the batch was created from -- options and constants the batch can follow in place go through
    `dmc_batch_set_opt_* / dmc_batch_set_model_real`, anything else rebuilds the batch from the edited model (the input
    state carried over), which is what `mj_step` seeing an edited mjModel amounts to.

That makes a `mj_step` through this seam cost a few host round trips: it is the drop-in path for code written against
`mujoco`, one environment at a time.  The throughput path is `BatchedPhysics` (thousands of environments per launch).

Not offered (AttributeError / NotImplementedError, never a silent no-op): rendering (`MjrContext`, `mjv_*`, `mjr_*`),
plugins, user callbacks other than `mjcb_passive / mjcb_control / mjcb_time`.
"""
import copy as _copy
import enum
import os
import pickle

import numpy as np

from dm_control_amd import _layout
from dm_control_amd import mjcf_compiler
from dm_control_amd.batch import BatchedPhysics      # (tests swap this name for the oracle stand-in)

C = mjcf_compiler.C

mjVERSION_HEADER = 3011000      # the version the reference pins (requirements.txt:9); this module restates its call surface
mjMAXVAL = C['DMC_MAXVAL']
mjMINVAL = C['DMC_MINVAL']
mjMINMU = C['DMC_MINMU']
mjMINIMP = C['DMC_MINIMP']
mjMAXIMP = C['DMC_MAXIMP']
mjPI = np.pi
mjNEQDATA = 11
mjNDYN = mjNGAIN = mjNBIAS = 10
mjNREF = 2
mjNIMP = 5
mjNSOLVER = 200
mjNISLAND = 20
mjMAXCONPAIR = 50


def mj_version():
  return mjVERSION_HEADER


def mj_versionString():
  return 'dm_control_amd HIP backend (mujoco 3.11.0 call surface)'


class FatalError(Exception):
  """mujoco.FatalError: what `mju_error` raises (migration_guide_1.0.md:47-58)."""


class UnexpectedError(Exception):
  pass


# ---------------------------------------------------------------------------------------------------------------------
# enums (values: include/dmc_model_layout.h where the device reads them, MuJoCo's documented order otherwise)
# ---------------------------------------------------------------------------------------------------------------------
def _enum(name, members):
  return enum.IntEnum(name, members, module=__name__)


def _seq(prefix, names, last=None):
  out = [(prefix + n, i) for i, n in enumerate(names)]
  if last:
    out.append((last, len(names)))
  return out


_DSBL_NAMES = ['CONSTRAINT', 'EQUALITY', 'FRICTIONLOSS', 'LIMIT', 'CONTACT', 'SPRING', 'DAMPER', 'GRAVITY', 'CLAMPCTRL',
               'WARMSTART', 'FILTERPARENT', 'ACTUATION', 'REFSAFE', 'SENSOR', 'MIDPHASE', 'EULERDAMP', 'AUTORESET',
               'NATIVECCD', 'ISLAND', 'MULTICCD']
mjtDisableBit = _enum('mjtDisableBit', [('mjDSBL_' + n, C['DMC_DSBL_' + n]) for n in _DSBL_NAMES] + [('mjNDISABLE', len(_DSBL_NAMES))])
_ENBL_NAMES = ['OVERRIDE', 'ENERGY', 'FWDINV', 'INVDISCRETE', 'SLEEP']
mjtEnableBit = _enum('mjtEnableBit', [('mjENBL_' + n, 1 << i) for i, n in enumerate(_ENBL_NAMES)] + [('mjNENABLE', len(_ENBL_NAMES))])
mjtJoint = _enum('mjtJoint', _seq('mjJNT_', ['FREE', 'BALL', 'SLIDE', 'HINGE']))
mjtGeom = _enum('mjtGeom', _seq('mjGEOM_', ['PLANE', 'HFIELD', 'SPHERE', 'CAPSULE', 'ELLIPSOID', 'CYLINDER', 'BOX', 'MESH', 'SDF'], 'mjNGEOMTYPES'))
mjtIntegrator = _enum('mjtIntegrator', _seq('mjINT_', ['EULER', 'RK4', 'IMPLICIT', 'IMPLICITFAST']))
mjtCone = _enum('mjtCone', _seq('mjCONE_', ['PYRAMIDAL', 'ELLIPTIC']))
mjtJacobian = _enum('mjtJacobian', _seq('mjJAC_', ['DENSE', 'SPARSE', 'AUTO']))
mjtSolver = _enum('mjtSolver', _seq('mjSOL_', ['PGS', 'CG', 'NEWTON']))
mjtTrn = _enum('mjtTrn', _seq('mjTRN_', ['JOINT', 'JOINTINPARENT', 'SLIDERCRANK', 'TENDON', 'SITE', 'BODY']) + [('mjTRN_UNDEFINED', 1000)])
mjtDyn = _enum('mjtDyn', _seq('mjDYN_', ['NONE', 'INTEGRATOR', 'FILTER', 'FILTEREXACT', 'MUSCLE', 'USER']))
mjtGain = _enum('mjtGain', _seq('mjGAIN_', ['FIXED', 'AFFINE', 'MUSCLE', 'USER']))
mjtBias = _enum('mjtBias', _seq('mjBIAS_', ['NONE', 'AFFINE', 'MUSCLE', 'USER']))
mjtEq = _enum('mjtEq', _seq('mjEQ_', ['CONNECT', 'WELD', 'JOINT', 'TENDON', 'FLEX', 'DISTANCE']))
mjtWrap = _enum('mjtWrap', _seq('mjWRAP_', ['NONE', 'JOINT', 'PULLEY', 'SITE', 'SPHERE', 'CYLINDER']))
mjtStage = _enum('mjtStage', _seq('mjSTAGE_', ['NONE', 'POS', 'VEL', 'ACC']))
mjtObj = _enum('mjtObj', _seq('mjOBJ_', ['UNKNOWN', 'BODY', 'XBODY', 'JOINT', 'DOF', 'GEOM', 'SITE', 'CAMERA', 'LIGHT', 'FLEX', 'MESH',
                                           'SKIN', 'HFIELD', 'TEXTURE', 'MATERIAL', 'PAIR', 'EXCLUDE', 'EQUALITY', 'TENDON',
                                           'ACTUATOR', 'SENSOR', 'NUMERIC', 'TEXT', 'TUPLE', 'KEY', 'PLUGIN'], 'mjNOBJECT')
                + [('mjOBJ_FRAME', 100)])
mjtSensor = _enum('mjtSensor', sorted([('mjSENS_' + k[len('DMC_SENS_'):], v) for k, v in C.items() if k.startswith('DMC_SENS_')],
                                      key=lambda kv: kv[1]))
# the device keeps one counter MuJoCo does not have (a geom pair its narrow phase cannot resolve came into range)
mjtWarning = _enum('mjtWarning', _seq('mjWARN_', ['INERTIA', 'CONTACTFULL', 'CNSTRFULL', 'VGEOMFULL', 'BADQPOS', 'BADQVEL', 'BADQACC', 'BADCTRL'])
                   + [('dmcWARN_COLLISION', 8), ('mjNWARNING', 9)])
mjtTimer = _enum('mjtTimer', _seq('mjTIMER_', ['STEP', 'FORWARD', 'INVERSE', 'POSITION', 'VELOCITY', 'ACTUATION', 'CONSTRAINT',
                                               'ADVANCE', 'POS_KINEMATICS', 'POS_INERTIA', 'POS_COLLISION', 'POS_MAKE',
                                               'POS_PROJECT', 'COL_BROAD', 'COL_NARROW'], 'mjNTIMER'))
mjtConstraint = _enum('mjtConstraint', _seq('mjCNSTR_', ['EQUALITY', 'FRICTION_DOF', 'FRICTION_TENDON', 'LIMIT_JOINT', 'LIMIT_TENDON',
                                                         'CONTACT_FRICTIONLESS', 'CONTACT_PYRAMIDAL', 'CONTACT_ELLIPTIC']))
_STATE_NAMES = ['TIME', 'QPOS', 'QVEL', 'ACT', 'WARMSTART', 'CTRL', 'QFRC_APPLIED', 'XFRC_APPLIED', 'EQ_ACTIVE', 'MOCAP_POS',
                'MOCAP_QUAT', 'USERDATA', 'PLUGIN']
_SB = {n: 1 << i for i, n in enumerate(_STATE_NAMES)}
_SB['PHYSICS'] = _SB['QPOS'] | _SB['QVEL'] | _SB['ACT']
_SB['FULLPHYSICS'] = _SB['TIME'] | _SB['PHYSICS'] | _SB['PLUGIN']
_SB['USER'] = (_SB['CTRL'] | _SB['QFRC_APPLIED'] | _SB['XFRC_APPLIED'] | _SB['EQ_ACTIVE'] | _SB['MOCAP_POS'] | _SB['MOCAP_QUAT'] | _SB['USERDATA'])
_SB['INTEGRATION'] = _SB['FULLPHYSICS'] | _SB['USER'] | _SB['WARMSTART']
mjtState = _enum('mjtState', [('mjSTATE_' + n, _SB[n]) for n in _STATE_NAMES] + [('mjNSTATE', len(_STATE_NAMES))]
                 + [('mjSTATE_' + n, _SB[n]) for n in ('PHYSICS', 'FULLPHYSICS', 'USER', 'INTEGRATION')])
# visualisation enums: values only (the reference's engine.py / core.py read them at import); nothing here renders
mjtFont = _enum('mjtFont', _seq('mjFONT_', ['NORMAL', 'SHADOW', 'BIG']))
mjtGridPos = _enum('mjtGridPos', _seq('mjGRID_', ['TOPLEFT', 'TOPRIGHT', 'BOTTOMLEFT', 'BOTTOMRIGHT', 'TOP', 'BOTTOM', 'LEFT', 'RIGHT']))
mjtFontScale = _enum('mjtFontScale', [('mjFONTSCALE_%d' % s, s) for s in (50, 100, 150, 200, 250, 300)])
mjtFramebuffer = _enum('mjtFramebuffer', [('mjFB_WINDOW', 0x10001), ('mjFB_OFFSCREEN', 0x10002)])
mjtCamera = _enum('mjtCamera', _seq('mjCAMERA_', ['FREE', 'TRACKING', 'FIXED', 'USER']))
mjtCatBit = _enum('mjtCatBit', [('mjCAT_STATIC', 1), ('mjCAT_DYNAMIC', 2), ('mjCAT_DECOR', 4), ('mjCAT_ALL', 7)])
mjtFrame = _enum('mjtFrame', _seq('mjFRAME_', ['NONE', 'BODY', 'GEOM', 'SITE', 'CAMERA', 'LIGHT', 'CONTACT', 'WORLD'], 'mjNFRAME'))
mjtLabel = _enum('mjtLabel', _seq('mjLABEL_', ['NONE', 'BODY', 'JOINT', 'GEOM', 'SITE', 'CAMERA', 'LIGHT', 'TENDON', 'ACTUATOR',
                                               'CONSTRAINT', 'FLEX', 'SKIN', 'SELECTION', 'SELPNT', 'CONTACTPOINT',
                                               'CONTACTFORCE', 'ISLAND'], 'mjNLABEL'))
mjtVisFlag = _enum('mjtVisFlag', _seq('mjVIS_', ['CONVEXHULL', 'TEXTURE', 'JOINT', 'CAMERA', 'ACTUATOR', 'ACTIVATION', 'LIGHT', 'TENDON',
                                                 'RANGEFINDER', 'CONSTRAINT', 'INERTIA', 'SCLINERTIA', 'PERTFORCE', 'PERTOBJ',
                                                 'CONTACTPOINT', 'ISLAND', 'CONTACTFORCE', 'CONTACTSPLIT', 'TRANSPARENT',
                                                 'AUTOCONNECT', 'COM', 'SELECT', 'STATIC', 'SKIN', 'FLEXVERT', 'FLEXEDGE',
                                                 'FLEXFACE', 'FLEXSKIN', 'BODYBVH', 'FLEXBVH', 'MESHBVH', 'SDFITER'], 'mjNVISFLAG'))
mjtRndFlag = _enum('mjtRndFlag', _seq('mjRND_', ['SHADOW', 'WIREFRAME', 'REFLECTION', 'ADDITIVE', 'SKYBOX', 'FOG', 'HAZE', 'SEGMENT',
                                                 'IDCOLOR', 'CULL_FACE'], 'mjNRNDFLAG'))

mjDISABLESTRING = tuple(n.capitalize() for n in _DSBL_NAMES)
mjENABLESTRING = tuple(n.capitalize() for n in _ENBL_NAMES)
mjTIMERSTRING = tuple(n[len('mjTIMER_'):].lower() for n in list(mjtTimer.__members__)[:-1])
mjLABELSTRING = tuple(n[len('mjLABEL_'):].capitalize() for n in list(mjtLabel.__members__)[:-1])
mjFRAMESTRING = tuple(n[len('mjFRAME_'):].capitalize() for n in list(mjtFrame.__members__)[:-1])
mjVISSTRING = tuple((n[len('mjVIS_'):].capitalize(), '0', '') for n in list(mjtVisFlag.__members__)[:-1])
mjRNDSTRING = tuple((n[len('mjRND_'):].capitalize(), '0', '') for n in list(mjtRndFlag.__members__)[:-1])

_TYPE_STRINGS = {'body': 1, 'xbody': 2, 'joint': 3, 'dof': 4, 'geom': 5, 'site': 6, 'camera': 7, 'light': 8, 'flex': 9, 'mesh': 10,
                 'skin': 11, 'hfield': 12, 'texture': 13, 'material': 14, 'pair': 15, 'exclude': 16, 'equality': 17, 'tendon': 18,
                 'actuator': 19, 'sensor': 20, 'numeric': 21, 'text': 22, 'tuple': 23, 'key': 24, 'plugin': 25}
# mjtObj -> the compiled model's name table
_OBJ_NAMES = {1: 'body', 2: 'body', 3: 'joint', 5: 'geom', 6: 'site', 7: 'camera', 8: 'light', 10: 'mesh', 12: 'hfield', 13: 'texture',
              14: 'material', 17: 'equality', 18: 'tendon', 19: 'actuator', 20: 'sensor', 21: 'numeric', 22: 'text', 23: 'tuple', 24: 'key'}


def mju_str2Type(s):
  s = s.decode() if isinstance(s, bytes) else s
  return _TYPE_STRINGS.get(s, 0)


def mju_type2Str(t):
  for k, v in _TYPE_STRINGS.items():
    if v == int(t):
      return k
  return None


# ---------------------------------------------------------------------------------------------------------------------
# process-wide callbacks (core.py:98-141): mjcb_passive / mjcb_control run on the host between launches, mjcb_time switches
# the launch timers on (core.enable_timer); the rest cannot reach a fused kernel
# ---------------------------------------------------------------------------------------------------------------------
_CALLBACK_NAMES = ('mjcb_passive', 'mjcb_control', 'mjcb_contactfilter', 'mjcb_sensor', 'mjcb_time', 'mjcb_act_dyn', 'mjcb_act_gain',
                   'mjcb_act_bias')
_HOST_CALLBACKS = ('mjcb_passive', 'mjcb_control', 'mjcb_time')
_callbacks = {n: None for n in _CALLBACK_NAMES}
_user_warning = [None]


def _make_cb(name):
  def setter(fn):
    if fn is not None and name not in _HOST_CALLBACKS:
      raise NotImplementedError('%s cannot be honoured: the physics step is one fused GPU kernel' % name)
    _callbacks[name] = fn

  def getter():
    return _callbacks[name]
  return setter, getter


for _n in _CALLBACK_NAMES:
  globals()['set_' + _n], globals()['get_' + _n] = _make_cb(_n)


def set_mju_user_warning(fn):
  _user_warning[0] = fn


def _warn(message):
  cb = _user_warning[0]
  if cb is None:
    return
  import ctypes
  cb(message.encode() if isinstance(cb, ctypes._CFuncPtr) else message)      # pylint: disable=protected-access  (core.py:62-74 registers a CFUNCTYPE(None, c_char_p))


def get_mju_user_warning():
  return _user_warning[0]


def set_mju_user_error(fn):
  del fn


# ---------------------------------------------------------------------------------------------------------------------
# small math (the mju_* helpers the reference calls through mjlib: utils/transformations.py, composer/entity.py)
# ---------------------------------------------------------------------------------------------------------------------
def mju_mulQuat(res, a, b):
  res[:] = mjcf_compiler.quat_mul(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))


def mju_negQuat(res, q):
  res[:] = np.asarray(q, dtype=np.float64) * np.array([1.0, -1, -1, -1])


def mju_rotVecQuat(res, vec, q):
  res[:] = mjcf_compiler.rot_vec(np.asarray(q, dtype=np.float64), np.asarray(vec, dtype=np.float64))


def mju_quat2Mat(res, q):
  np.asarray(res).reshape(-1)[:] = mjcf_compiler.quat_to_mat(np.asarray(q, dtype=np.float64)).ravel()


def mju_mat2Quat(res, mat):
  res[:] = mjcf_compiler.mat_to_quat(np.asarray(mat, dtype=np.float64).reshape(3, 3))


def mju_axisAngle2Quat(res, axis, angle):
  res[0] = np.cos(angle / 2)
  res[1:4] = np.asarray(axis, dtype=np.float64) * np.sin(angle / 2)


def mju_quat2Vel(res, quat, dt):
  q = np.asarray(quat, dtype=np.float64)
  axis = q[1:4].copy()
  sin_a_2 = np.linalg.norm(axis)
  if sin_a_2 > 0:
    axis /= sin_a_2
  speed = 2 * np.arctan2(sin_a_2, q[0])
  if speed > np.pi:
    speed -= 2 * np.pi
  res[:] = axis * speed / dt


def mju_sym2dense(res, mat, rownnz, rowadr, colind):
  """Dense symmetric matrix from the lower-triangle CSR `mat` (mjData.M with mjModel.M_rownnz / M_rowadr / M_colind)."""
  res = np.asarray(res)
  res[...] = 0
  mat, rownnz, rowadr, colind = (np.asarray(a).ravel() for a in (mat, rownnz, rowadr, colind))
  for i in range(res.shape[0]):
    for k in range(int(rowadr[i]), int(rowadr[i] + rownnz[i])):
      j = int(colind[k])
      res[i, j] = res[j, i] = mat[k]


def mj_fullM(m, dst, qM):
  """Dense mass matrix from the legacy sparse `qM` (ancestor chains, diagonal first)."""
  dst = np.asarray(dst)
  dst[...] = 0
  x = _extras(m)
  for (i, j), v in zip(x['qM_ij'], np.asarray(qM).ravel()):
    dst[i, j] = dst[j, i] = v


# ---------------------------------------------------------------------------------------------------------------------
# MjModel
# ---------------------------------------------------------------------------------------------------------------------
class _Global:
  offwidth, offheight = 640, 480
  fovy = 45.0


class _Vis:
  def __init__(self):
    self.global_ = _Global()


class _Stat:
  def __init__(self, compiled):
    self._c = compiled

  @property
  def meaninertia(self):
    return float(self._c.stat_meaninertia)

  @meaninertia.setter
  def meaninertia(self, v):
    self._c.stat_meaninertia = float(v)


_ROW_ORDER = ('body', 'joint', 'geom', 'site', 'camera', 'light', 'mesh', 'hfield', 'texture', 'material', 'equality', 'tendon',
              'actuator', 'sensor', 'numeric', 'text', 'tuple', 'key')
_ADR_FIELD = {'body': 'name_bodyadr', 'joint': 'name_jntadr', 'geom': 'name_geomadr', 'site': 'name_siteadr', 'camera': 'name_camadr',
              'light': 'name_lightadr', 'mesh': 'name_meshadr', 'hfield': 'name_hfieldadr', 'texture': 'name_texadr',
              'material': 'name_matadr', 'equality': 'name_eqadr', 'tendon': 'name_tendonadr', 'actuator': 'name_actuatoradr',
              'sensor': 'name_sensoradr', 'numeric': 'name_numericadr', 'text': 'name_textadr', 'tuple': 'name_tupleadr',
              'key': 'name_keyadr'}
_COUNT_FIELD = {'body': 'nbody', 'joint': 'njnt', 'geom': 'ngeom', 'site': 'nsite', 'camera': 'ncam', 'light': 'nlight', 'mesh': 'nmesh',
                'hfield': 'nhfield', 'texture': 'ntex', 'material': 'nmat', 'equality': 'neq', 'tendon': 'ntendon', 'actuator': 'nu',
                'sensor': 'nsensor', 'numeric': 'nnumeric', 'text': 'ntext', 'tuple': 'ntuple', 'key': 'nkey'}


def _world_frames_at_qpos0(c):
  """xpos / xquat / subtree_com at qpos0 (hinge and slide displacements zero, ball quaternions as stored): what the
  `*0` camera / light constants of mjModel are measured in."""
  nb = c.nbody
  xpos, xquat = np.zeros((nb, 3)), np.tile([1.0, 0, 0, 0], (nb, 1))
  for b in range(1, nb):
    p, ja, jn = int(c.body_parentid[b]), int(c.body_jntadr[b]), int(c.body_jntnum[b])
    if jn == 1 and c.jnt_type[ja] == 0:
      a = int(c.jnt_qposadr[ja])
      xpos[b] = c.qpos0[a:a + 3]
      xquat[b] = c.qpos0[a + 3:a + 7] / np.linalg.norm(c.qpos0[a + 3:a + 7])
    else:
      xpos[b] = xpos[p] + mjcf_compiler.rot_vec(xquat[p], c.body_pos[b])
      xquat[b] = mjcf_compiler.quat_mul(xquat[p], c.body_quat[b])
      for j in range(ja, ja + jn):
        if c.jnt_type[j] == 1:
          a = int(c.jnt_qposadr[j])
          xquat[b] = mjcf_compiler.quat_mul(xquat[b], c.qpos0[a:a + 4] / np.linalg.norm(c.qpos0[a:a + 4]))
  xipos = np.array([xpos[b] + mjcf_compiler.rot_vec(xquat[b], c.body_ipos[b]) for b in range(nb)]).reshape(nb, 3)
  com = xipos * np.asarray(c.body_mass, dtype=np.float64)[:, None]
  mass = np.asarray(c.body_mass, dtype=np.float64).copy()
  for b in range(nb - 1, 0, -1):
    com[c.body_parentid[b]] += com[b]
    mass[c.body_parentid[b]] += mass[b]
  com = np.where(mass[:, None] > 0, com / np.maximum(mass[:, None], mjMINVAL), xipos)
  return xpos, xquat, com


def _build_extras(c):
  """What mjModel holds beyond the compiled tables: the name buffer and its address arrays, the CSR pattern of M, the
  activation addresses, camera / light constants at qpos0, dof ancestry used by the host-side derivations."""
  x = {}
  buf = bytearray((c.model_name or '').encode() + b'\0')
  for kind in _ROW_ORDER:
    adr = []
    for nm in c.names.get(kind, [None] * int(getattr(c, _COUNT_FIELD[kind], 0))):
      adr.append(len(buf))
      buf += (nm or '').encode() + b'\0'
    x[_ADR_FIELD[kind]] = np.array(adr, dtype=np.int32)
  x['names'] = bytes(buf)
  x['nnames'] = len(buf)
  nv = c.nv
  parent = np.asarray(c.dof_parentid, dtype=np.int64)
  rows, qM_ij = [], []
  for i in range(nv):
    chain, j = [], i
    while j >= 0:
      chain.append(j)
      j = int(parent[j])
    rows.append(sorted(chain))
    qM_ij.extend((i, j) for j in chain)       # legacy qM: the diagonal first, then up the chain
  x['M_rownnz'] = np.array([len(r) for r in rows], dtype=np.int32)
  x['M_rowadr'] = np.concatenate([[0], np.cumsum(x['M_rownnz'])])[:-1].astype(np.int32)
  x['M_colind'] = np.array([j for r in rows for j in r], dtype=np.int32)
  x['nC'] = x['nM'] = int(x['M_colind'].size)
  x['qM_ij'] = qM_ij
  x['dof_Madr'] = np.concatenate([[0], np.cumsum(x['M_rownnz'])])[:-1].astype(np.int32)
  dyn = np.asarray(c.actuator_dyntype, dtype=np.int64)
  x['actuator_actnum'] = (dyn != 0).astype(np.int32)
  x['actuator_actadr'] = np.where(dyn != 0, np.cumsum(dyn != 0) - 1, -1).astype(np.int32)
  # body c is moved by dof k iff k's body is c or an ancestor of c
  anc = np.zeros((c.nbody, nv), dtype=bool)
  for b in range(1, c.nbody):
    anc[b] = anc[int(c.body_parentid[b])]
    d0, dn = int(c.body_dofadr[b]), int(c.body_dofnum[b])
    if dn:
      anc[b, d0:d0 + dn] = True
  x['body_dofmask'] = anc
  sub = np.zeros((c.nbody, c.nbody), dtype=bool)      # sub[r, b]: b is in the subtree rooted at r
  for b in range(c.nbody):
    a = b
    while True:
      sub[a, b] = True
      if a == 0:
        break
      a = int(c.body_parentid[a])
  x['subtree'] = sub
  xpos0, xquat0, com0 = _world_frames_at_qpos0(c)
  for pre, n, bodyid, pos in (('cam', c.ncam, getattr(c, 'cam_bodyid', np.zeros(0, int)), getattr(c, 'cam_pos', np.zeros((0, 3)))),
                              ('light', c.nlight, c.light_bodyid, c.light_pos)):
    gp = np.array([xpos0[bodyid[i]] + mjcf_compiler.rot_vec(xquat0[bodyid[i]], pos[i]) for i in range(n)]).reshape(n, 3)
    x[pre + '_pos0'] = gp - xpos0[bodyid].reshape(n, 3)
    x[pre + '_poscom0'] = gp - com0[bodyid].reshape(n, 3)
  x['cam_mat0'] = np.array([mjcf_compiler.quat_to_mat(mjcf_compiler.quat_mul(xquat0[c.cam_bodyid[i]], c.cam_quat[i])).ravel()
                            for i in range(c.ncam)]).reshape(c.ncam, 9)
  x['light_dir0'] = np.array([mjcf_compiler.rot_vec(xquat0[c.light_bodyid[i]], c.light_dir[i]) for i in range(c.nlight)]).reshape(c.nlight, 3)
  x['mesh_normal'] = np.zeros((0, 3))
  x['skin_rgba'] = np.zeros((0, 4))
  x['tendon_rgba'] = np.tile([0.5, 0.5, 0.5, 1.0], (c.ntendon, 1)).reshape(c.ntendon, 4)
  return x


_EXTRA_ARRAYS = ('name_bodyadr', 'name_jntadr', 'name_geomadr', 'name_siteadr', 'name_camadr', 'name_lightadr', 'name_meshadr',
                 'name_hfieldadr', 'name_texadr', 'name_matadr', 'name_eqadr', 'name_tendonadr', 'name_actuatoradr', 'name_sensoradr',
                 'name_numericadr', 'name_textadr', 'name_tupleadr', 'name_keyadr', 'M_rownnz', 'M_rowadr', 'M_colind', 'dof_Madr',
                 'actuator_actnum', 'actuator_actadr', 'cam_pos0', 'cam_poscom0', 'cam_mat0', 'light_pos0', 'light_poscom0',
                 'light_dir0', 'mesh_normal', 'skin_rgba', 'tendon_rgba')
_EXTRA_SCALARS = ('names', 'nnames', 'nC', 'nM')
# sizes of mjModel beyond the compiled model's own
_ZERO_SIZES = ('nflex', 'nskin', 'nplugin', 'npluginstate', 'nuserdata', 'nuser_body', 'nuser_jnt', 'nuser_geom', 'nuser_site',
               'nuser_cam', 'nuser_tendon', 'nuser_actuator', 'nuser_sensor', 'nexclude', 'nemax')


def _extras(m):
  c = m._c if isinstance(m, MjModel) else m
  x = c.__dict__.get('_mj_extras')
  if x is None:
    x = _build_extras(c)
    c.__dict__['_mj_extras'] = x
  return x


class MjOption:
  """mjModel.opt: attribute access onto the compiled model's options (`timestep gravity integrator disableflags ...`)."""

  def __init__(self, compiled):
    object.__setattr__(self, '_c', compiled)

  def __getattr__(self, name):
    o = self._c.opt
    if not hasattr(o, name):
      raise AttributeError(name)
    return getattr(o, name)

  def __setattr__(self, name, value):
    o = self._c.opt
    if not hasattr(o, name):
      raise AttributeError(name)
    cur = getattr(o, name)
    if isinstance(cur, np.ndarray):
      cur[...] = value
    elif isinstance(cur, float):
      setattr(o, name, float(value))
    else:
      setattr(o, name, int(value))

  def __dir__(self):
    return sorted(vars(self._c.opt))

  def __eq__(self, other):
    return isinstance(other, MjOption) and all(np.array_equal(getattr(self, k), getattr(other, k)) for k in dir(self))


def _model_attr_names():
  probe = mjcf_compiler.compile_xml(
      "<mujoco><worldbody><light name='l'/><camera name='c'/><body name='b'><joint name='j'/><geom name='g' size='.1'/>"
      "<site name='s'/></body></worldbody><actuator><motor name='a' joint='j'/></actuator>"
      "<sensor><jointpos name='p' joint='j'/></sensor></mujoco>")
  arrays, scalars = [], []
  for name, value in vars(probe).items():
    if name.startswith('_') or name in ('opt', 'names', 'model_name', 'exclude_bodies', 'stat_meaninertia'):
      continue
    (arrays if isinstance(value, np.ndarray) else scalars).append(name)
  return arrays, scalars


class _ModelMeta(type):

  def __new__(mcs, name, bases, dct):
    arrays, scalars = _model_attr_names()

    def compiled_attr(attr):
      def fget(self):
        v = getattr(self._c, attr)
        return int(v) if isinstance(v, (int, np.integer)) and not isinstance(v, bool) else v

      def fset(self, value):
        cur = getattr(self._c, attr)
        if isinstance(cur, np.ndarray):
          cur[...] = value
        else:
          setattr(self._c, attr, type(cur)(value))
      return property(fget, fset)

    def extra_attr(attr):
      def fget(self):
        return _extras(self)[attr]

      def fset(self, value):
        cur = _extras(self)[attr]
        if isinstance(cur, np.ndarray):
          cur[...] = value
        else:
          raise AttributeError('%s is read-only' % attr)
      return property(fget, fset)
    for a in arrays + scalars:
      dct.setdefault(a, compiled_attr(a))
    for a in _EXTRA_ARRAYS + _EXTRA_SCALARS:
      dct.setdefault(a, extra_attr(a))
    for a in _ZERO_SIZES:
      dct.setdefault(a, property(lambda self: 0))
    dct['_ARRAYS'] = tuple(arrays) + _EXTRA_ARRAYS
    return super().__new__(mcs, name, bases, dct)


_last_xml = {}      # id(compiled) -> (xml string, assets): what mj_saveLastXML writes back


class MjModel(metaclass=_ModelMeta):
  """mujoco.MjModel: the compiled constant tables, every array writable (see the module docstring)."""

  def __init__(self, *args, **kwargs):
    raise TypeError('MjModel cannot be constructed directly; use MjModel.from_xml_string / from_xml_path / from_binary_path')

  @classmethod
  def _wrap(cls, compiled):
    self = object.__new__(cls)
    self._c = compiled
    return self

  @classmethod
  def from_xml_string(cls, xml, assets=None):
    xml = xml.decode() if isinstance(xml, bytes) else xml
    self = cls._wrap(mjcf_compiler.compile_xml(xml, assets))      # (MjcfError is a ValueError: what a failed load raises)
    self._c.__dict__['_mj_xml'] = (xml, dict(assets or {}))
    return self

  @classmethod
  def from_xml_path(cls, filename, assets=None):
    try:
      with open(filename) as f:
        xml = f.read()
    except OSError as e:
      raise ValueError('could not open %r: %s' % (filename, e))
    merged = dict(assets or {})
    base = os.path.dirname(os.path.abspath(filename))
    return cls.from_xml_string(xml, _DirAssets(base, merged))

  @classmethod
  def from_binary_path(cls, filename, assets=None):
    data = (assets or {}).get(filename)
    if data is None:
      try:
        with open(filename, 'rb') as f:
          data = f.read()
      except OSError as e:
        raise ValueError('could not open %r: %s' % (filename, e))
    return cls._wrap(_load_mjb(bytes(data)))

  # -- the MuJoCo bindings' own conveniences ----------------------------------------------------------------------
  @property
  def opt(self):
    return MjOption(self._c)

  @property
  def vis(self):
    return self._c.__dict__.setdefault('_mj_vis', _Vis())

  @property
  def stat(self):
    return _Stat(self._c)

  @property
  def njmax(self):
    return -1

  @property
  def nconmax(self):
    return -1

  @property
  def nmocap(self):
    return int(getattr(self._c, 'nmocap', 0))

  @property
  def nmeshvert(self):
    return int(self._c.mesh_vert.shape[0])

  @property
  def nsensordata(self):
    return int(self._c.nsensordata)

  def __copy__(self):
    c = _copy.deepcopy(self._c)
    c.__dict__.pop('_mj_extras', None)
    return MjModel._wrap(c)

  def __deepcopy__(self, memo):
    return self.__copy__()

  def __getstate__(self):
    return {'c': self._c}

  def __setstate__(self, st):
    self._c = st['c']

  def __reduce__(self):
    return (_unpickle_model, (pickle.dumps(self._c, protocol=4),))


def _unpickle_model(blob):
  return MjModel._wrap(pickle.loads(blob))


class _DirAssets(dict):
  """Assets of a model loaded from a path: files named by the model resolve relative to its directory."""

  def __init__(self, base, given):
    super().__init__(given)
    self._base = base

  def _path(self, key):
    return key if os.path.isabs(key) else os.path.join(self._base, key)

  def __contains__(self, key):
    return dict.__contains__(self, key) or os.path.isfile(self._path(key))

  def __getitem__(self, key):
    if dict.__contains__(self, key):
      return dict.__getitem__(self, key)
    with open(self._path(key), 'rb') as f:
      return f.read()

  def get(self, key, default=None):
    return self[key] if key in self else default

  def __reduce__(self):
    return (dict, (dict(self),))


_MJB_MAGIC = b'DMCMJB01'


def _dump_mjb(m):
  c = m._c
  keep = {k: v for k, v in c.__dict__.items() if k not in ('_mj_extras', '_mj_vis', '_mj_xml')}
  clone = object.__new__(type(c))
  clone.__dict__.update(keep)
  return _MJB_MAGIC + pickle.dumps(clone, protocol=4)


def _load_mjb(data):
  if not data.startswith(_MJB_MAGIC):
    raise ValueError('not a model binary written by this backend')
  return pickle.loads(data[len(_MJB_MAGIC):])


def mj_sizeModel(m):
  return len(_dump_mjb(m))


def mj_saveModel(m, filename=None, buffer=None):
  blob = _dump_mjb(m)
  if filename:
    with open(filename, 'wb') as f:
      f.write(blob)
  if buffer is not None:
    buf = np.asarray(buffer).reshape(-1).view(np.uint8)
    if buf.size < len(blob):
      raise ValueError('buffer too small: %d < %d' % (buf.size, len(blob)))
    buf[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)


def mj_saveLastXML(filename, m):
  """Writes the XML the model was parsed from with the model's CURRENT real-valued frames copied back into it (MuJoCo's
  mj_copyBack): `pos / quat / size` of bodies, geoms and sites, joint `pos / axis`."""
  import xml.etree.ElementTree as ET
  src = m._c.__dict__.get('_mj_xml')
  if src is None:
    raise FatalError('mj_saveLastXML: the model was not parsed from XML')
  comp = mjcf_compiler._Compiler(src[0], src[1])      # pylint: disable=protected-access  (includes expanded, elements in id order)
  root, c = comp.root, m._c
  fmt = lambda a: ' '.join(repr(float(v)) for v in np.asarray(a).ravel())
  drop = ('euler', 'axisangle', 'xyaxes', 'zaxis', 'fromto')
  bodies, geoms, sites, joints = [], [], [], []

  # the compiler numbers a body's own elements before descending: geoms / sites / joints in document order per body,
  # bodies depth first -- the same order walk() visits them in when children are handled after the body's own elements
  def ordered(e):
    own = [ch for ch in e if ch.tag != 'body']
    for ch in own:
      if ch.tag == 'geom':
        geoms.append(ch)
      elif ch.tag == 'site':
        sites.append(ch)
      elif ch.tag in ('joint', 'freejoint'):
        joints.append(ch)
    for ch in e:
      if ch.tag == 'body':
        bodies.append(ch)
        ordered(ch)
  wbs = root.findall('worldbody')
  for wb in wbs:
    ordered(wb)
  if len(geoms) == c.ngeom:
    for g, e in enumerate(geoms):
      if c.geom_type[g] == C['DMC_GEOM_MESH']:
        continue
      for k in drop:
        e.attrib.pop(k, None)
      e.set('pos', fmt(c.geom_pos[g]))
      e.set('quat', fmt(c.geom_quat[g]))
      n = {0: 3, 2: 1, 3: 2, 4: 3, 5: 2, 6: 3}.get(int(c.geom_type[g]), 3)
      e.set('size', fmt(c.geom_size[g][:n]))
  if len(bodies) == c.nbody - 1:
    for b, e in enumerate(bodies, start=1):
      for k in drop:
        e.attrib.pop(k, None)
      e.set('pos', fmt(c.body_pos[b]))
      e.set('quat', fmt(c.body_quat[b]))
  if len(sites) == c.nsite:
    for s, e in enumerate(sites):
      for k in drop:
        e.attrib.pop(k, None)
      e.set('pos', fmt(c.site_pos[s]))
      e.set('quat', fmt(c.site_quat[s]))
  with open(filename, 'w') as f:
    f.write(ET.tostring(root, encoding='unicode'))


def mj_printSchema(filename, buffer, buffer_sz, flg_html, flg_pad):
  del filename, flg_html, flg_pad
  text = b'<mujoco> (schema: dm_control/mjcf/schema.xml; this backend compiles the subset listed in DESIGN.md)'
  n = min(len(text), int(buffer_sz) - 1)
  try:
    buffer[:n] = text[:n]
  except TypeError:
    for i in range(n):
      buffer[i] = text[i:i + 1]
  return n


def mj_name2id(m, type_, name):
  kind = _OBJ_NAMES.get(int(type_))
  name = name.decode() if isinstance(name, bytes) else name
  if kind is None or not name:
    return -1
  lst = m._c.names.get(kind, [])
  return lst.index(name) if name in lst else -1


def mj_id2name(m, type_, id_):
  kind = _OBJ_NAMES.get(int(type_))
  if kind is None:
    return None
  lst = m._c.names.get(kind, [])
  return (lst[id_] or None) if 0 <= id_ < len(lst) else None


# ---------------------------------------------------------------------------------------------------------------------
# MjData
# ---------------------------------------------------------------------------------------------------------------------
_IN = ('qpos', 'qvel', 'act', 'ctrl', 'qacc_warmstart', 'qfrc_applied', 'xfrc_applied', 'mocap_pos', 'mocap_quat')
_OUT = ('sensordata', 'xpos', 'xquat', 'xmat', 'xipos', 'geom_xpos', 'geom_xmat', 'site_xpos', 'site_xmat', 'subtree_com', 'qacc',
        'actuator_force', 'qfrc_actuator', 'qfrc_bias', 'qfrc_constraint', 'cvel')
_DERIVED = ('ximat', 'xanchor', 'xaxis', 'cam_xpos', 'cam_xmat', 'light_xpos', 'light_xdir', 'ten_length', 'ten_velocity', 'wrap_xpos',
            'M', 'qM', 'qLD', 'qLDiagInv', 'subtree_linvel', 'subtree_angmom', 'act_dot', 'energy', 'qfrc_passive')
_ON_REQUEST = ('subtree_linvel', 'subtree_angmom')
# field -> shape as (size name | int, ...): what mjbindings.sizes.array_sizes lists for mjData
DATA_SHAPES = {
    'qpos': ('nq',), 'qvel': ('nv',), 'act': ('na',), 'ctrl': ('nu',), 'qacc_warmstart': ('nv',), 'qfrc_applied': ('nv',),
    'xfrc_applied': ('nbody', 6), 'mocap_pos': ('nmocap', 3), 'mocap_quat': ('nmocap', 4), 'eq_active': ('neq',),
    'sensordata': ('nsensordata',), 'xpos': ('nbody', 3), 'xquat': ('nbody', 4), 'xmat': ('nbody', 9), 'xipos': ('nbody', 3),
    'geom_xpos': ('ngeom', 3), 'geom_xmat': ('ngeom', 9), 'site_xpos': ('nsite', 3), 'site_xmat': ('nsite', 9),
    'subtree_com': ('nbody', 3), 'qacc': ('nv',), 'actuator_force': ('nu',), 'qfrc_actuator': ('nv',), 'qfrc_bias': ('nv',),
    'qfrc_constraint': ('nv',), 'cvel': ('nbody', 6),
    'ximat': ('nbody', 9), 'xanchor': ('njnt', 3), 'xaxis': ('njnt', 3), 'cam_xpos': ('ncam', 3), 'cam_xmat': ('ncam', 9),
    'light_xpos': ('nlight', 3), 'light_xdir': ('nlight', 3), 'ten_length': ('ntendon',), 'ten_velocity': ('ntendon',),
    'wrap_xpos': ('nwrap', 6), 'M': ('nC',), 'qM': ('nM',), 'qLD': ('nC',), 'qLDiagInv': ('nv',), 'subtree_linvel': ('nbody', 3),
    'subtree_angmom': ('nbody', 3), 'act_dot': ('na',), 'qfrc_passive': ('nv',),
}
_CONTACT_DTYPE = np.dtype([('dist', np.float64), ('pos', np.float64, 3), ('frame', np.float64, 9), ('includemargin', np.float64),
                           ('friction', np.float64, 5), ('solref', np.float64, 2), ('solreffriction', np.float64, 2),
                           ('solimp', np.float64, 5), ('mu', np.float64), ('H', np.float64, 36), ('dim', np.int32),
                           ('geom1', np.int32), ('geom2', np.int32), ('geom', np.int32, 2), ('flex', np.int32, 2),
                           ('elem', np.int32, 2), ('vert', np.int32, 2), ('exclude', np.int32), ('efc_address', np.int32)])
_AUTO_NCONMAX = (64, 48, 32, 0)      # as the Physics facade: generous contact capacity first, whatever fits in LDS


def _size(m, s):
  if isinstance(s, int):
    return s
  c = m._c
  if s in ('nC', 'nM'):
    return _extras(c)[s]
  return int(getattr(c, s, 0))


class _StatList:
  """mjData.warning / mjData.timer: a struct array whose members are arrays (`.number`) and whose items have them as
  scalars (`warning[k].number = 1`)."""

  def __init__(self, fields):
    self._fields = fields      # name -> ndarray

  def __getattr__(self, name):
    f = self.__dict__.get('_fields', {})
    if name in f:
      return f[name]
    raise AttributeError(name)

  def __len__(self):
    return len(next(iter(self._fields.values())))

  def __getitem__(self, k):
    return _StatItem(self._fields, int(k))

  def __iter__(self):
    return (self[k] for k in range(len(self)))

  def __eq__(self, other):
    return (isinstance(other, _StatList) and set(self._fields) == set(other._fields)
            and all(np.array_equal(v, other._fields[k]) for k, v in self._fields.items()))

  __hash__ = None

  def __repr__(self):
    return '_StatList(%s)' % ', '.join('%s=%r' % kv for kv in self._fields.items())


class _StatItem:

  def __init__(self, fields, k):
    object.__setattr__(self, '_f', fields)
    object.__setattr__(self, '_k', k)

  def __getattr__(self, name):
    if name in self._f:
      return self._f[name][self._k].item()
    raise AttributeError(name)

  def __setattr__(self, name, value):
    if name not in self._f:
      raise AttributeError(name)
    self._f[name][self._k] = value


class _TimerList:
  """mjData.timer: [mjTIMER_STEP] and [mjTIMER_FORWARD] are the hipEvent brackets around the launches
  (dmc_batch_enable_profiling), live once `mjcb_time` is set (core.enable_timer); the fused launch has no sub-stages."""

  def __init__(self, data):
    self._d = data

  def __len__(self):
    return int(mjtTimer.mjNTIMER)

  def __getitem__(self, k):
    k = int(k)
    d = self._d

    class T:
      @property
      def duration(self):
        return d._batch.timer(k)[0] if k < 2 and hasattr(d._batch, 'timer') else 0.0

      @property
      def number(self):
        return d._batch.timer(k)[1] if k < 2 and hasattr(d._batch, 'timer') else 0
    return T()

  def __eq__(self, other):
    return isinstance(other, _TimerList)

  __hash__ = None


def _data_property(name):
  def fget(self):
    return self._array(name)

  def fset(self, value):
    self._array(name)[...] = value
  return property(fget, fset)


class _DataMeta(type):

  def __new__(mcs, name, bases, dct):
    for f in DATA_SHAPES:
      dct.setdefault(f, _data_property(f))
    dct.setdefault('energy', _data_property('energy'))
    return super().__new__(mcs, name, bases, dct)


class MjData(metaclass=_DataMeta):
  """mujoco.MjData(model): one environment of a batch-size-1 `dmc_batch` plus the host arrays described above."""

  def __init__(self, model):
    if not isinstance(model, MjModel):
      raise TypeError('MjData(model): expected an MjModel, got %r' % type(model).__name__)
    self._model = model
    self._arrays = {}
    self._shadow = {}            # input field -> what the device holds
    self._time = 0.0
    self._time_dev = 0.0
    self._outputs_valid = False  # mj_makeData / mj_resetData leave every derived array zero until something is computed
    self._fresh_forward = False  # the device's derived arrays belong to an mj_forward at exactly the current inputs
    self._warn_dev = np.zeros(int(mjtWarning.mjNWARNING), dtype=np.int64)
    self._warning = _StatList({'number': np.zeros(int(mjtWarning.mjNWARNING), dtype=np.int32),
                               'lastinfo': np.zeros(int(mjtWarning.mjNWARNING), dtype=np.int32)})
    self._solver = _StatList({k: np.zeros(mjNSOLVER) for k in ('improvement', 'gradient', 'lineslope')}
                             | {k: np.zeros(mjNSOLVER, dtype=np.int32) for k in ('nactive', 'nchange', 'neval', 'nupdate')})
    self._ints = {'ncon': 0, 'nefc': 0, 'solver_niter': 0}
    self._batch = None
    self._make_batch()
    self._pull_inputs()

  # -- the batch ------------------------------------------------------------------------------------------------
  def _packed_model(self):
    c = self._model._c
    if 'eq_active' in self._arrays and c.neq:
      saved = c.eq_active0
      c.eq_active0 = np.asarray(self._arrays['eq_active'], dtype=np.int64)
      try:
        return c.pack()
      finally:
        c.eq_active0 = saved
    return c.pack()

  def _make_batch(self, carry=None):
    c = self._model._c
    mjcf_compiler.candidate_pairs(c)      # (geom_contype / geom_conaffinity may have been edited)
    if 'eq_active' in self._arrays and c.neq:
      src = _copy.copy(c)
      src.eq_active0 = np.asarray(self._arrays['eq_active'], dtype=np.int64)
    else:
      src = c
    err = None
    for cap in _AUTO_NCONMAX:
      try:
        batch = BatchedPhysics(src, 1, device_id=0, precision=64, nconmax=cap)
        break
      except Exception as e:      # pylint: disable=broad-except
        err = e
        if cap == 0 or 'does not fit' not in str(e):
          raise
    else:
      raise err
    old, self._batch = self._batch, batch
    self._nconmax = int(batch.info().get('nconmax', 0)) or 16
    self._pushed = tuple(a.copy() for a in self._packed_model())
    self._fingerprint = self._model_fingerprint()
    if carry is not None:
      for name, a in carry.items():
        if name == 'xfrc_applied' and not a.any():
          continue
        batch.set(name, a.reshape(1, -1))
    if old is not None and hasattr(old, 'close'):
      old.close()
    if _callbacks['mjcb_time'] is not None and hasattr(batch, 'enable_profiling'):
      batch.enable_profiling(True)
      self._profiling = True

  def _pull_inputs(self):
    """Host input arrays := the device's (after creation, reset or a launch)."""
    for name in _IN:
      dev = np.asarray(self._batch.get(name), dtype=np.float64).reshape(self._shape(name))
      self._shadow[name] = dev
      if name in self._arrays:
        np.copyto(self._arrays[name], dev)
    self._time = self._time_dev = float(np.asarray(self._batch.get('time')).ravel()[0])

  def _shape(self, name):
    if name == 'energy':
      return (2,)
    return tuple(_size(self._model, s) for s in DATA_SHAPES[name])

  def _array(self, name):
    a = self._arrays.get(name)
    if a is None:
      shape = self._shape(name)
      if name in _IN:
        a = np.array(self._shadow[name], dtype=np.float64).reshape(shape)
      elif name == 'eq_active':
        a = np.array(self._model._c.eq_active0, dtype=np.uint8).reshape(shape)
      else:
        a = np.zeros(shape, dtype=np.float64)
        self._arrays[name] = a
        if self._outputs_valid:
          self._refresh_fields([name])
      self._arrays[name] = a
    return a

  # -- scalars and structs ------------------------------------------------------------------------------------------
  @property
  def model(self):
    return self._model

  @property
  def time(self):
    return self._time

  @time.setter
  def time(self, v):
    self._time = float(v)

  @property
  def ncon(self):
    return self._ints['ncon']

  @property
  def nefc(self):
    return self._ints['nefc']

  @property
  def solver_niter(self):
    return np.array([self._ints['solver_niter']], dtype=np.int32)

  @property
  def warning(self):
    return self._warning

  @property
  def solver(self):
    return self._solver

  @property
  def timer(self):
    return _TimerList(self)

  @property
  def contact(self):
    """The active contacts: a record array of length ncon (a view of one persistent buffer of the contact capacity)."""
    buf = self._arrays.get('contact')
    if buf is None:
      buf = np.zeros(self._nconmax, dtype=_CONTACT_DTYPE).view(np.recarray)
      self._arrays['contact'] = buf
      if self._outputs_valid:
        self._refresh_fields(['contact'])
    return buf[:self.ncon]

  # -- synchronisation ------------------------------------------------------------------------------------------------
  _OPT_INT = ('disableflags', 'iterations', 'ls_iterations', 'noslip_iterations')
  _OPT_REAL = {'opt_timestep': 'timestep', 'opt_gravity_x': 'gravity_x', 'opt_gravity_y': 'gravity_y', 'opt_gravity_z': 'gravity_z',
               'opt_tolerance': 'tolerance', 'opt_ls_tolerance': 'ls_tolerance', 'opt_noslip_tolerance': 'noslip_tolerance'}
  _MUTABLE = ('dof_damping', 'jnt_stiffness', 'jnt_range', 'jnt_margin', 'qpos_spring', 'site_pos', 'site_quat', 'site_size',
              'actuator_ctrlrange', 'actuator_forcerange', 'wrap_prm', 'body_pos', 'body_quat', 'geom_pos', 'geom_quat', 'geom_size')

  def _model_fingerprint(self):
    """Every number the model blob is made of, flat (one concatenate: this runs before every launch)."""
    c = self._model._c
    x = _extras(c)
    names = x.get('_blob_fields')
    if names is None:
      names = x['_blob_fields'] = [n for n, _ in _layout.INT_FIELDS] + [n for n, _ in _layout.REAL_FIELDS]
    o = c.opt
    parts = [np.asarray(getattr(c, n), dtype=np.float64).ravel() for n in names]
    parts.append(np.array([o.timestep, o.gravity[0], o.gravity[1], o.gravity[2], o.impratio, o.tolerance, o.ls_tolerance,
                           o.noslip_tolerance, o.density, o.viscosity, c.stat_meaninertia, o.integrator, o.cone, o.solver,
                           o.iterations, o.ls_iterations, o.noslip_iterations, o.disableflags, o.enableflags], dtype=np.float64))
    ea = self._arrays.get('eq_active')
    if ea is not None:
      parts.append(np.asarray(ea, dtype=np.float64))
    return np.concatenate(parts)

  def _sync_model(self):
    """Brings the device's model tables up to the MjModel as it stands now (see the module docstring)."""
    fp = self._model_fingerprint()
    old = self.__dict__.get('_fingerprint')
    if old is not None and old.shape == fp.shape and np.array_equal(fp, old, equal_nan=True):
      return
    self._fingerprint = fp
    ints, reals = self._packed_model()
    pi, pr = self._pushed
    if ints.shape == pi.shape and reals.shape == pr.shape and np.array_equal(ints, pi) and np.array_equal(reals, pr, equal_nan=True):
      return
    c = self._model._c
    rebuild = ints.shape != pi.shape or reals.shape != pr.shape
    if not rebuild:
      sizes = c.sizes()
      # header ints: [magic, version, sizes..., options...]
      hdr = 2 + len(_layout.HEADER_INTS)
      di = np.nonzero(ints != pi)[0]
      for k in di:
        if k >= hdr:
          rebuild = True
          break
        name = _layout.HEADER_INTS[k - 2]
        if not name.startswith('opt_') or name[4:] not in self._OPT_INT:
          rebuild = True
          break
    if not rebuild:
      for k in di:
        self._batch.set_opt(_layout.HEADER_INTS[k - 2][4:], int(ints[k]))
      nh = len(_layout.HEADER_REALS)
      same = (reals == pr) | (np.isnan(reals) & np.isnan(pr))
      for k in np.nonzero(~same[:nh])[0]:
        name = _layout.HEADER_REALS[k]
        if name not in self._OPT_REAL:
          rebuild = True
          break
      off, touched = nh, []
      if not rebuild:
        for name, expr in _layout.REAL_FIELDS:
          n = _layout.field_count(expr, sizes)
          if n and not same[off:off + n].all():
            if name not in self._MUTABLE:
              rebuild = True
              break
            touched.append(name)
          off += n
    if rebuild:
      carry = {n: np.asarray(self._batch.get(n), dtype=np.float64) for n in _IN}
      carry['time'] = np.asarray(self._batch.get('time'), dtype=np.float64)
      self._make_batch(carry)
      self._fresh_forward = False
      return
    for k in np.nonzero(~same[:nh])[0]:
      self._batch.set_opt(self._OPT_REAL[_layout.HEADER_REALS[k]], float(reals[k]))
    for name in touched:
      self._batch.set_model_real(name, np.asarray(getattr(c, name), dtype=np.float64))
    self._pushed = (ints.copy(), reals.copy())
    self._fresh_forward = False

  def _upload(self):
    """Input arrays that differ from what the device holds go up."""
    self._sync_model()
    changed = False
    for name in _IN:
      a = self._arrays.get(name)
      if a is None:
        continue
      dev = self._shadow[name]
      if not np.array_equal(a, dev, equal_nan=True):
        if name == 'xfrc_applied' and not a.any() and not getattr(self, '_xfrc_sent', False):
          continue
        if name == 'xfrc_applied':
          self._xfrc_sent = True
        self._batch.set(name, a.reshape(1, -1))
        self._shadow[name] = a.copy()
        changed = True
    if self._time != self._time_dev:
      self._batch.set('time', np.array([[self._time]]))
      self._time_dev = self._time
      changed = True
    ea = self._arrays.get('eq_active')
    if ea is not None and self._model._c.neq:
      pass      # (part of the packed model: _sync_model rebuilt the batch if it changed)
    if changed:
      self._fresh_forward = False

  def _refresh_fields(self, names):
    """Rewrites the handed-out arrays `names` in place from the device (and the host-side derivations)."""
    b, A = self._batch, self._arrays
    dev = [n for n in names if n in _OUT]
    got = {}
    if dev:
      if hasattr(b, 'get_many'):
        for k in range(0, len(dev), 8):
          got.update(b.get_many(dev[k:k + 8]))
      else:
        got = {n: b.get(n) for n in dev}
      for n in dev:
        np.copyto(A[n], np.asarray(got[n], dtype=np.float64).reshape(A[n].shape))
    if 'contact' in names:
      buf = A['contact']
      buf[...] = np.zeros((), dtype=_CONTACT_DTYPE)
      n = self.ncon
      if n:
        buf['geom1'][:n] = np.asarray(b.get('contact_geom1')).ravel()[:n]
        buf['geom2'][:n] = np.asarray(b.get('contact_geom2')).ravel()[:n]
        buf['geom'][:n, 0], buf['geom'][:n, 1] = buf['geom1'][:n], buf['geom2'][:n]
        buf['dist'][:n] = np.asarray(b.get('contact_dist')).ravel()[:n]
        buf['pos'][:n] = np.asarray(b.get('contact_pos')).reshape(-1, 3)[:n]
        buf['frame'][:n] = np.asarray(b.get('contact_frame')).reshape(-1, 9)[:n]
        c = self._model._c
        buf['dim'][:n] = np.maximum(c.geom_condim[buf['geom1'][:n]], c.geom_condim[buf['geom2'][:n]])
        buf['exclude'][:n] = 0
        buf['efc_address'][:n] = -1
    der = [n for n in names if n in _DERIVED]
    if der:
      _derive(self, der)

  def _after_launch(self, state_changed):
    b = self._batch
    self._outputs_valid = True
    for k in ('ncon', 'nefc'):
      self._ints[k] = int(np.asarray(b.get(k)).ravel()[0])
    self._ints['solver_niter'] = int(np.asarray(b.get('solver_iter')).ravel()[0])
    w = np.asarray(b.get('warning')).astype(np.int64).ravel()
    new = w[:self._warn_dev.size] - self._warn_dev[:w.size]
    self._warn_dev[:w.size] = w[:self._warn_dev.size]
    if new.any():
      self._warning.number[:new.size] += new.astype(np.int32)
      for k in np.nonzero(new > 0)[0]:
        _warn('%s (the device counted it %d time(s)). Time = %.4f.' % (list(mjtWarning.__members__)[k], int(new[k]), self._time))
    if state_changed:
      self._pull_inputs()
    elif new[int(mjtWarning.mjWARN_BADCTRL)] > 0:
      # mj_fwdActuation zeroes its COPY of a bad control vector: mjData.ctrl keeps what the caller wrote, the device's
      # array does not -- the next launch uploads (and reports) it again
      self._shadow['ctrl'] = np.asarray(b.get('ctrl'), dtype=np.float64).reshape(self._shape('ctrl'))
    # (subtree_linvel / subtree_angmom: mj_subtreeVel is not part of mj_step / mj_forward -- as in MuJoCo they are brought up
    # by an explicit mj_subtreeVel call, which legacy_base.Walker.after_substep makes)
    self._refresh_fields([n for n in self._arrays if n in _OUT or (n in _DERIVED and n not in _ON_REQUEST) or n == 'contact'])

  def _host_callbacks(self):
    """mjcb_control / mjcb_passive: called on the host before the launch of ONE physics step.  Whatever mjcb_passive adds
    to qfrc_passive enters that step as an applied force (both are summed into qfrc_smooth)."""
    extra = None
    cb = _callbacks['mjcb_control']
    if cb is not None:
      cb(self._model, self)
    cb = _callbacks['mjcb_passive']
    if cb is not None:
      qp = self._array('qfrc_passive')
      before = qp.copy()
      cb(self._model, self)
      if not np.array_equal(qp, before):
        extra = qp - before
    return extra

  def _launch(self, kind, nstep=1):
    if _callbacks['mjcb_time'] is not None and not getattr(self, '_profiling', False) and hasattr(self._batch, 'enable_profiling'):
      self._batch.enable_profiling(True)
      self._profiling = True
    hostcb = _callbacks['mjcb_control'] is not None or _callbacks['mjcb_passive'] is not None
    if hostcb and kind == 'step' and nstep > 1:
      for _ in range(nstep):
        self._launch('step', 1)
      return
    extra = self._host_callbacks() if hostcb and kind in ('step', 'step2', 'forward') else None
    if extra is not None:
      fa = self._array('qfrc_applied')
      keep = fa.copy()
      fa += extra
    self._upload()
    b = self._batch
    integ = int(self._model._c.opt.integrator)
    if kind == 'step':
      b.legacy_step = False
      b.step(int(nstep))
    elif kind == 'step1':
      b.step1()
    elif kind == 'step2':
      if integ == C['DMC_INT_RK4']:
        raise FatalError('mj_step2 with the RK4 integrator is not offered by this backend (engine.py:154-160 never asks for it)')
      b.step2()
    elif kind == 'forward':
      b.forward(bool(int(self._model._c.opt.disableflags) & C['DMC_DSBL_ACTUATION']))
    self._fresh_forward = kind == 'forward' and extra is None
    self._after_launch(kind in ('step', 'step2'))
    if extra is not None:
      fa[:] = keep      # (differs from what the device now holds: the next launch uploads it)

  def _ensure_forward(self):
    """mj_fwdActuation / mj_fwdAcceleration / mj_fwdConstraint: the fused kernel cannot run one stage of mj_forward
    alone, so each of them is mj_forward at the current inputs -- run once, with the solver warm start put back (a query
    must not move it)."""
    self._upload()
    if self._fresh_forward:
      return
    warm = np.asarray(self._batch.get('qacc_warmstart'), dtype=np.float64)
    self._launch('forward')
    self._batch.set('qacc_warmstart', warm)
    self._shadow['qacc_warmstart'] = warm.reshape(self._shape('qacc_warmstart')).copy()
    if 'qacc_warmstart' in self._arrays:
      np.copyto(self._arrays['qacc_warmstart'], self._shadow['qacc_warmstart'])
    self._fresh_forward = True

  # -- copy / pickle ------------------------------------------------------------------------------------------------
  def _snapshot(self):
    """Everything a copy needs: the inputs, and every derived array AS IT STANDS (after mj_step the position-dependent
    arrays belong to the state before the integration; a copy must show the same values, so they are carried, not
    recomputed)."""
    self._upload()
    arrays = {k: np.array(v) for k, v in self._arrays.items()}
    if self._outputs_valid:
      for n in _OUT:
        if n not in arrays:
          arrays[n] = np.asarray(self._batch.get(n), dtype=np.float64).reshape(self._shape(n))
      if 'contact' not in arrays:
        self.contact      # pylint: disable=pointless-statement  (materialises the buffer)
        arrays['contact'] = np.array(self._arrays['contact'])
    return {'inputs': {n: np.asarray(self._batch.get(n), dtype=np.float64) for n in _IN},
            'time': self._time, 'arrays': arrays,
            'ints': dict(self._ints), 'warning': {k: v.copy() for k, v in self._warning._fields.items()},
            'solver': {k: v.copy() for k, v in self._solver._fields.items()},
            'outputs_valid': self._outputs_valid}

  def _restore(self, snap):
    b = self._batch
    for n, a in snap['inputs'].items():
      if n == 'xfrc_applied' and not a.any():
        continue
      if n == 'xfrc_applied':
        self._xfrc_sent = True
      b.set(n, a.reshape(1, -1))
    b.set('time', np.array([[snap['time']]]))
    self._pull_inputs()
    self._fresh_forward = False
    self._ints = dict(snap['ints'])
    for k, v in snap['warning'].items():
      self._warning._fields[k][...] = v
    for k, v in snap['solver'].items():
      self._solver._fields[k][...] = v
    self._outputs_valid = False      # (materialise without reading THIS batch's device arrays: they hold nothing yet)
    for k, v in snap['arrays'].items():
      if k == 'contact':
        self.contact      # pylint: disable=pointless-statement  (allocates the buffer)
        n = min(len(v), len(self._arrays['contact']))
        self._arrays['contact'][:n] = v[:n]
      else:
        np.copyto(self._array(k), v)
    self._outputs_valid = snap['outputs_valid']

  def __copy__(self):
    other = MjData(self._model)
    other._restore(self._snapshot())
    return other

  def __deepcopy__(self, memo):
    other = MjData(_copy.copy(self._model))
    other._restore(self._snapshot())
    return other

  def __reduce__(self):
    return (_unpickle_data, (self._model, self._snapshot()))

  def __del__(self):
    try:
      b = self.__dict__.get('_batch')
      if b is not None and hasattr(b, 'close'):
        b.close()
    except Exception:      # pylint: disable=broad-except
      pass


def _unpickle_data(model, snap):
  d = MjData(model)
  d._restore(snap)
  return d


# ---------------------------------------------------------------------------------------------------------------------
# host-side derivations of the mjData arrays the kernel keeps on chip
# ---------------------------------------------------------------------------------------------------------------------
def _cross(a, b):
  return np.stack([a[..., 1]*b[..., 2] - a[..., 2]*b[..., 1], a[..., 2]*b[..., 0] - a[..., 0]*b[..., 2],
                   a[..., 0]*b[..., 1] - a[..., 1]*b[..., 0]], axis=-1)


def _joint_plan(c):
  """Per model: the joints grouped by their rank counted from the LAST joint of their body (pass r of joint_frames
  handles every rank-r joint of the model at once)."""
  x = _extras(c)
  plan = x.get('_joint_plan')
  if plan is None:
    rank = np.zeros(c.njnt, dtype=np.int64)
    for b in range(c.nbody):
      j0, jn = int(c.body_jntadr[b]), int(c.body_jntnum[b])
      for k in range(jn):
        rank[j0 + k] = jn - 1 - k
    typ = np.asarray(c.jnt_type, dtype=np.int64)
    passes = []
    for r in range(int(rank.max()) + 1 if c.njnt else 0):
      js = np.nonzero(rank == r)[0]
      passes.append({'j': js, 'b': np.asarray(c.jnt_bodyid, dtype=np.int64)[js], 't': typ[js],
                     'qa': np.asarray(c.jnt_qposadr, dtype=np.int64)[js]})
    plan = x['_joint_plan'] = passes
  return plan


def joint_frames(c, qpos, xpos, xquat, mocap_pos=None, mocap_quat=None):
  """mjData.xanchor / xaxis.  mj_kinematics takes each joint's anchor and axis in the body frame accumulated BEFORE that
  joint moves it; here the walk runs the other way, from the body's FINAL frame (the device's xpos / xquat) back through
  its joints: a hinge or ball rotation leaves its own anchor and axis where they were, a slide moves the frame along its
  axis.  One vectorised pass per joint rank within a body (at most three in the suite models)."""
  del mocap_pos, mocap_quat      # (a mocap body has no joints; its children start from its device frame like any other)
  anchor, axis = np.zeros((c.njnt, 3)), np.zeros((c.njnt, 3))
  if not c.njnt:
    return anchor, axis
  pos, quat = np.array(xpos, dtype=np.float64), np.array(xquat, dtype=np.float64)      # current frame per body
  jaxis, jpos = np.asarray(c.jnt_axis, dtype=np.float64), np.asarray(c.jnt_pos, dtype=np.float64)
  q0 = np.asarray(c.qpos0, dtype=np.float64)
  for ps in _joint_plan(c):
    js, bs, ts, qa = ps['j'], ps['b'], ps['t'], ps['qa']
    R = _quat_to_mat_rows(quat[bs]).reshape(-1, 3, 3)
    ax = np.einsum('nij,nj->ni', R, jaxis[js])
    an = pos[bs] + np.einsum('nij,nj->ni', R, jpos[js])
    free, ball, slide, hinge = ts == 0, ts == 1, ts == 2, ts == 3
    if free.any():
      k = np.nonzero(free)[0]
      an[k] = np.stack([qpos[qa[k]], qpos[qa[k] + 1], qpos[qa[k] + 2]], axis=1)
      ax[k] = jaxis[js[k]]
    if slide.any():
      k = np.nonzero(slide)[0]
      shift = ax[k] * (qpos[qa[k]] - q0[qa[k]])[:, None]
      an[k] -= shift
      pos[bs[k]] -= shift
    rot = hinge | ball
    if rot.any():
      k = np.nonzero(rot)[0]
      qloc = np.zeros((k.size, 4))
      kh = hinge[k]
      if kh.any():
        ang = (qpos[qa[k[kh]]] - q0[qa[k[kh]]]) / 2
        qloc[kh, 0] = np.cos(ang)
        qloc[kh, 1:] = jaxis[js[k[kh]]] * np.sin(ang)[:, None]
      kb = ~kh
      if kb.any():
        qb = np.stack([qpos[qa[k[kb]] + i] for i in range(4)], axis=1)
        qloc[kb] = qb / np.linalg.norm(qb, axis=1, keepdims=True)
      qprev = _quat_mul_rows(quat[bs[k]], qloc * np.array([1.0, -1, -1, -1]))
      quat[bs[k]] = qprev
      Rp = _quat_to_mat_rows(qprev).reshape(-1, 3, 3)
      pos[bs[k]] = an[k] - np.einsum('nij,nj->ni', Rp, jpos[js[k]])
      if kb.any():      # (a ball joint turns its own nominal axis: mjData.xaxis is the axis BEFORE the joint acts)
        ax[k[kb]] = np.einsum('nij,nj->ni', Rp[kb], jaxis[js[k[kb]]])
    anchor[js], axis[js] = an, ax
  return anchor, axis


def _quat_mul_rows(a, b):
  w1, x1, y1, z1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
  w2, x2, y2, z2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
  return np.stack([w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2,
                   w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2], axis=1)


def _quat_to_mat_rows(q):
  w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
  return np.stack([w*w + x*x - y*y - z*z, 2*(x*y - w*z), 2*(x*z + w*y),
                   2*(x*y + w*z), w*w - x*x + y*y - z*z, 2*(y*z - w*x),
                   2*(x*z - w*y), 2*(y*z + w*x), w*w - x*x - y*y + z*z], axis=1)


def _dof_plan(c):
  x = _extras(c)
  plan = x.get('_dof_plan')
  if plan is None:
    nv = c.nv
    kind = np.zeros(nv, dtype=np.int64)      # 0: world axis (free translation), 1: a column of the body's xmat, 2: the joint's xaxis
    col, jnt, body, rot, fixed_anchor = (np.zeros(nv, dtype=np.int64) for _ in range(5))
    for j in range(c.njnt):
      d, t, b = int(c.jnt_dofadr[j]), int(c.jnt_type[j]), int(c.jnt_bodyid[j])
      n = {0: 6, 1: 3}.get(t, 1)
      jnt[d:d + n], body[d:d + n] = j, b
      if t == 0:
        kind[d:d + 3], col[d:d + 3] = 0, np.arange(3)
        kind[d + 3:d + 6], col[d + 3:d + 6], rot[d + 3:d + 6], fixed_anchor[d + 3:d + 6] = 1, np.arange(3), 1, 1
      elif t == 1:
        kind[d:d + 3], col[d:d + 3], rot[d:d + 3] = 1, np.arange(3), 1
      else:
        kind[d], rot[d] = 2, int(t == 3)
    plan = x['_dof_plan'] = dict(kind=kind, col=col, jnt=jnt, body=body, rot=rot.astype(bool), free_rot=fixed_anchor.astype(bool))
  return plan


def mass_matrix(c, xpos, xmat, xipos, ximat, xanchor, xaxis):
  """Dense joint-space inertia M(q) (what mj_crb leaves in mjData.M) from world-frame body Jacobians:
  M = sum_b m_b Jp_b' Jp_b + Jr_b' (R_b I_b R_b') Jr_b + diag(dof_armature)."""
  nv, nb = c.nv, c.nbody
  x = _extras(c)
  p = _dof_plan(c)
  R = np.asarray(xmat, dtype=np.float64).reshape(nb, 3, 3)
  axis = np.where((p['kind'] == 2)[:, None], xaxis[p['jnt']], R[p['body'], :, p['col']])
  axis = np.where((p['kind'] == 0)[:, None], np.eye(3)[p['col']], axis)
  anchor = np.where(p['free_rot'][:, None], xpos[p['body']], xanchor[p['jnt']])
  rot = p['rot']
  mask = x['body_dofmask']                                   # (nb, nv)
  r = xipos[:, None, :] - anchor[None, :, :]                 # (nb, nv, 3)
  jp = np.where(rot[None, :, None], _cross(np.broadcast_to(axis[None, :, :], r.shape), r), axis[None, :, :]) * mask[:, :, None]
  jr = np.where(rot[None, :, None], axis[None, :, :], 0.0) * mask[:, :, None]
  Ri = np.asarray(ximat, dtype=np.float64).reshape(nb, 3, 3)
  jl = np.einsum('bji,bkj->bki', Ri, jr)                     # angular Jacobian in the inertial frame: (nb, nv, 3)
  M = np.einsum('b,bki,bli->kl', np.asarray(c.body_mass, dtype=np.float64), jp, jp)
  M += np.einsum('bki,bi,bli->kl', jl, np.asarray(c.body_inertia, dtype=np.float64), jl)
  M[np.diag_indices(nv)] += np.asarray(c.dof_armature, dtype=np.float64)
  return M


def _derive(d, names):
  c, A = d._model._c, d._arrays
  x = _extras(c)
  need = set(names)
  b = d._batch
  get = lambda n, *shape: np.asarray(b.get(n), dtype=np.float64).reshape(shape)

  def dev(n, *shape):      # a device array: the handed-out copy if there is one (it was just refreshed)
    return A[n].reshape(shape) if n in A and n in _OUT else get(n, *shape)
  nb = c.nbody
  heavy = need & {'M', 'qM', 'qLD', 'qLDiagInv', 'energy'}
  frames = need & {'xanchor', 'xaxis'} or heavy
  xpos = xquat = None
  if frames or need & {'ximat', 'cam_xpos', 'cam_xmat', 'light_xpos', 'light_xdir', 'subtree_linvel', 'subtree_angmom'}:
    xpos, xquat = dev('xpos', nb, 3), dev('xquat', nb, 4)
  ximat = None
  if 'ximat' in need or heavy or 'subtree_angmom' in need:
    ximat = _quat_to_mat_rows(_quat_mul_rows(xquat, np.asarray(c.body_iquat, dtype=np.float64).reshape(nb, 4)))
    if 'ximat' in A:
      np.copyto(A['ximat'], ximat)
  if frames:
    nm = int(getattr(c, 'nmocap', 0))
    anchor, axis = joint_frames(c, d._shadow['qpos'].ravel(), xpos, xquat,
                                d._shadow['mocap_pos'].reshape(nm, 3) if nm else None,
                                d._shadow['mocap_quat'].reshape(nm, 4) if nm else None)
    if 'xanchor' in A:
      np.copyto(A['xanchor'], anchor)
    if 'xaxis' in A:
      np.copyto(A['xaxis'], axis)
  for pre, n, bodyid, pos in (('cam', c.ncam, getattr(c, 'cam_bodyid', None), getattr(c, 'cam_pos', None)),
                              ('light', c.nlight, c.light_bodyid, c.light_pos)):
    if pre + '_xpos' in need and n:
      R = _quat_to_mat_rows(xquat[bodyid]).reshape(n, 3, 3)
      np.copyto(A[pre + '_xpos'], xpos[bodyid] + np.einsum('nij,nj->ni', R, pos))
  if 'cam_xmat' in need and c.ncam:
    np.copyto(A['cam_xmat'], _quat_to_mat_rows(_quat_mul_rows(xquat[c.cam_bodyid], np.asarray(c.cam_quat, dtype=np.float64))))
  if 'light_xdir' in need and c.nlight:
    R = _quat_to_mat_rows(xquat[c.light_bodyid]).reshape(c.nlight, 3, 3)
    np.copyto(A['light_xdir'], np.einsum('nij,nj->ni', R, c.light_dir))
  if need & {'ten_length', 'ten_velocity', 'wrap_xpos'} and c.ntendon:
    _tendons(d, dev)
  if 'act_dot' in need and c.na:
    _act_dot(d)
  if 'qfrc_passive' in need and c.nv:
    _passive(d, dev)
  if need & {'subtree_linvel', 'subtree_angmom'}:
    _subtree_vel(d, dev, xpos, ximat)
  if heavy and c.nv:
    xmat, xipos = dev('xmat', nb, 9), dev('xipos', nb, 3)
    M = mass_matrix(c, xpos, xmat, xipos, ximat, anchor, axis)
    if 'M' in A:
      rows = np.repeat(np.arange(c.nv), x['M_rownnz'])
      np.copyto(A['M'], M[rows, x['M_colind']])
    if 'qM' in A:
      ij = np.asarray(x['qM_ij'], dtype=np.int64).reshape(-1, 2)
      np.copyto(A['qM'], M[ij[:, 0], ij[:, 1]])
    if 'qLD' in A or 'qLDiagInv' in A:
      # M = L' D L, L unit lower triangular (mj_factorM): from the Cholesky factor of the index-reversed matrix
      G = np.linalg.cholesky(M[::-1, ::-1])
      U = G[::-1, ::-1]                      # upper triangular, M = U U'
      dg = np.diag(U)
      L = (U / dg[None, :]).T
      if 'qLD' in A:
        rows = np.repeat(np.arange(c.nv), x['M_rownnz'])
        q = L[rows, x['M_colind']]
        q[x['M_rowadr'] + x['M_rownnz'] - 1] = dg * dg
        np.copyto(A['qLD'], q)
      if 'qLDiagInv' in A:
        np.copyto(A['qLDiagInv'], 1.0 / (dg * dg))
    if 'energy' in A:
      e = A['energy']
      e[:] = 0
      if int(c.opt.enableflags) & C['DMC_ENBL_ENERGY']:
        qv = d._shadow['qvel'].ravel()
        g = np.asarray(c.opt.gravity, dtype=np.float64)
        pot = 0.0
        if not int(c.opt.disableflags) & C['DMC_DSBL_GRAVITY']:
          pot -= float(np.sum(np.asarray(c.body_mass)[1:, None] * xipos[1:] * g[None, :]))
        qp = d._shadow['qpos'].ravel()
        for j in range(c.njnt):
          if c.jnt_stiffness[j] and c.jnt_type[j] in (2, 3):
            a = int(c.jnt_qposadr[j])
            pot += 0.5 * c.jnt_stiffness[j] * (qp[a] - c.qpos_spring[a]) ** 2
        e[0], e[1] = pot, 0.5 * qv @ M @ qv
  elif 'energy' in need and 'energy' in A:
    A['energy'][:] = 0


def _tendons(d, dev):
  c, A = d._model._c, d._arrays
  nt = c.ntendon
  length, velocity = np.zeros(nt), np.zeros(nt)
  wx = np.zeros((c.nwrap, 6))
  qpos, qvel = d._shadow['qpos'].ravel(), d._shadow['qvel'].ravel()
  spatial = [t for t in range(nt) if c.tendon_num[t] and c.wrap_type[c.tendon_adr[t]] != C['DMC_WRAP_JOINT']]
  if spatial:
    sx, cvel, com = dev('site_xpos', c.nsite, 3), dev('cvel', c.nbody, 6), dev('subtree_com', c.nbody, 3)
  for t in range(nt):
    w0, wn = int(c.tendon_adr[t]), int(c.tendon_num[t])
    if t not in spatial:
      for w in range(w0, w0 + wn):
        j = int(c.wrap_objid[w])
        length[t] += c.wrap_prm[w] * qpos[c.jnt_qposadr[j]]
        velocity[t] += c.wrap_prm[w] * qvel[c.jnt_dofadr[j]]
      continue

    def point(w):
      sid = int(c.wrap_objid[w])
      bd = int(c.site_bodyid[sid])
      p = sx[sid]
      return p, cvel[bd, 3:] + np.cross(cvel[bd, :3], p - com[c.body_rootid[bd]])
    for w in range(w0, w0 + wn - 1):
      (p0, v0), (p1, v1) = point(w), point(w + 1)
      wx[w, :3], wx[w, 3:] = p0, p1
      dif = p1 - p0
      n = np.linalg.norm(dif)
      length[t] += n
      if n > mjMINVAL:
        velocity[t] += (dif / n) @ (v1 - v0)
  for name, val in (('ten_length', length), ('ten_velocity', velocity), ('wrap_xpos', wx)):
    if name in A:
      np.copyto(A[name], val)


def _passive(d, dev):
  """mjData.qfrc_passive (mj_passive): joint springs and dampers, fixed-tendon springs and dampers.  Fluid forces and
  ball / free joint springs are computed on the device only: a model that has them is refused here rather than served a
  partial sum."""
  c, A = d._model._c, d._arrays
  if float(c.opt.density) or float(c.opt.viscosity):
    raise NotImplementedError('mjData.qfrc_passive of a model with fluid forces is not derived on the host')
  qpos, qvel = d._shadow['qpos'].ravel(), d._shadow['qvel'].ravel()
  flags = int(c.opt.disableflags)
  out = np.zeros(c.nv)
  if not flags & C['DMC_DSBL_DAMPER']:
    out -= np.asarray(c.dof_damping, dtype=np.float64) * qvel
  if not flags & C['DMC_DSBL_SPRING']:
    for j in range(c.njnt):
      k = float(c.jnt_stiffness[j])
      if not k:
        continue
      if c.jnt_type[j] not in (2, 3):
        raise NotImplementedError('mjData.qfrc_passive with a spring on a ball / free joint is not derived on the host')
      out[c.jnt_dofadr[j]] -= k * (qpos[c.jnt_qposadr[j]] - c.qpos_spring[c.jnt_qposadr[j]])
  for t in range(c.ntendon):
    ks, kd = float(c.tendon_stiffness[t]), float(c.tendon_damping[t])
    if not (ks or kd):
      continue
    w0, wn = int(c.tendon_adr[t]), int(c.tendon_num[t])
    js = [int(c.wrap_objid[w]) for w in range(w0, w0 + wn)]
    coef = np.asarray(c.wrap_prm[w0:w0 + wn], dtype=np.float64)
    length = float(coef @ qpos[np.asarray(c.jnt_qposadr)[js]])
    vel = float(coef @ qvel[np.asarray(c.jnt_dofadr)[js]])
    f = 0.0
    if ks and not flags & C['DMC_DSBL_SPRING']:
      f -= ks * (length - float(c.tendon_lengthspring[t]))
    if kd and not flags & C['DMC_DSBL_DAMPER']:
      f -= kd * vel
    out[np.asarray(c.jnt_dofadr)[js]] += coef * f
  np.copyto(A['qfrc_passive'], out)


def _act_dot(d):
  """mjData.act_dot (mj_fwdActuation): integrator `ctrl`, filter `(ctrl - act) / tau`; filterexact has the same rate."""
  c, A = d._model._c, d._arrays
  ctrl, act = d._shadow['ctrl'].ravel(), d._shadow['act'].ravel()
  adr = _extras(c)['actuator_actadr']
  out = np.zeros(c.na)
  for i in range(c.nu):
    if adr[i] < 0:
      continue
    u = ctrl[i]
    if c.actuator_ctrllimited[i] and not int(c.opt.disableflags) & C['DMC_DSBL_CLAMPCTRL']:
      u = min(max(u, c.actuator_ctrlrange[i, 0]), c.actuator_ctrlrange[i, 1])
    t = int(c.actuator_dyntype[i])
    if t == C['DMC_DYN_INTEGRATOR']:
      out[adr[i]] = u
    else:
      out[adr[i]] = (u - act[adr[i]]) / max(mjMINVAL, c.actuator_dynprm[i, 0])
  if int(c.opt.disableflags) & C['DMC_DSBL_ACTUATION']:
    out[:] = 0
  np.copyto(A['act_dot'], out)


def _subtree_vel(d, dev, xpos, ximat):
  """mj_subtreeVel: linear velocity of every subtree's centre of mass and its angular momentum about it, from the
  device's com-based body velocities."""
  del xpos
  c, A = d._model._c, d._arrays
  nb = c.nbody
  cvel, com, xipos = dev('cvel', nb, 6), dev('subtree_com', nb, 3), dev('xipos', nb, 3)
  mass = np.asarray(c.body_mass, dtype=np.float64)
  sub = _extras(c)['subtree'].astype(np.float64)                  # (root, body)
  ang = cvel[:, :3]
  lin = cvel[:, 3:] + _cross(ang, xipos - com[np.asarray(c.body_rootid)])      # velocity of each body's own COM
  msub = np.maximum(sub @ mass, mjMINVAL)
  vsub = (sub @ (mass[:, None] * lin)) / msub[:, None]
  if 'subtree_linvel' in A:
    np.copyto(A['subtree_linvel'], vsub)
  if 'subtree_angmom' in A:
    R = ximat.reshape(nb, 3, 3)
    spin = np.einsum('bij,bj,bkj,bk->bi', R, np.asarray(c.body_inertia, dtype=np.float64), R, ang)
    # sum_b [I w + m (x - X) x (v - V)] = sum_b [I w + m x x v] - M X x V   (X, V: the subtree's centre of mass and its velocity)
    own = spin + mass[:, None] * _cross(xipos, lin)
    np.copyto(A['subtree_angmom'], sub @ own - msub[:, None] * _cross(com, vsub))


# ---------------------------------------------------------------------------------------------------------------------
# the engine calls
# ---------------------------------------------------------------------------------------------------------------------
def _check(m, d):
  if not isinstance(m, MjModel) or not isinstance(d, MjData):
    raise TypeError('expected (MjModel, MjData), got (%s, %s)' % (type(m).__name__, type(d).__name__))
  if d._model._c is not m._c:
    raise ValueError('this MjData was made for a different MjModel')


def mj_step(m, d, nstep=1):
  """engine.py:158,160,176."""
  _check(m, d)
  d._launch('step', int(nstep))


def mj_step1(m, d):
  """engine.py:162: position / velocity stage, position and velocity sensors, state unchanged."""
  _check(m, d)
  d._launch('step1')


def mj_step2(m, d):
  """engine.py:156: actuation, acceleration, constraint solve, integration."""
  _check(m, d)
  d._launch('step2')


def mj_forward(m, d):
  """engine.py:343."""
  _check(m, d)
  d._launch('forward')


def mj_fwdActuation(m, d):
  _check(m, d)
  d._ensure_forward()


mj_fwdAcceleration = mj_fwdConstraint = mj_fwdPosition = mj_fwdVelocity = mj_sensorPos = mj_sensorVel = mj_sensorAcc = mj_fwdActuation


def mj_kinematics(m, d):
  _check(m, d)
  d._ensure_forward()


mj_comPos = mj_comVel = mj_tendon = mj_kinematics


def mj_subtreeVel(m, d):
  """legacy_base.py:148,186: the device serves subtree velocities to its sensors itself; the mjData arrays are derived on
  the host whenever they have been handed out."""
  _check(m, d)
  if d._outputs_valid:
    d._refresh_fields([n for n in ('subtree_linvel', 'subtree_angmom') if n in d._arrays])


def _reset_host(d):
  d._pull_inputs()
  d._outputs_valid = False
  d._fresh_forward = False
  for name, a in d._arrays.items():
    if name in _IN:
      continue
    if name == 'eq_active':
      a[...] = np.asarray(d._model._c.eq_active0, dtype=np.uint8)
    elif name == 'contact':
      a[...] = np.zeros((), dtype=_CONTACT_DTYPE)
    else:
      a[...] = 0
  d._ints = {'ncon': 0, 'nefc': 0, 'solver_niter': 0}
  d._warning.number[:] = 0
  d._warning.lastinfo[:] = 0
  w = np.asarray(d._batch.get('warning')).astype(np.int64).ravel()
  d._warn_dev[:] = 0
  d._warn_dev[:w.size] = w[:d._warn_dev.size]


def mj_resetData(m, d):
  """engine.py:318."""
  _check(m, d)
  d._sync_model()
  d._batch.reset()
  _reset_host(d)


def mj_resetDataKeyframe(m, d, key):
  """engine.py:323."""
  _check(m, d)
  if not 0 <= int(key) < m._c.nkey:
    return mj_resetData(m, d)
  d._sync_model()
  d._batch.reset(keyframe_id=int(key))
  _reset_host(d)


def _state_parts(m, sig):
  sig = int(sig)
  if sig < 0 or sig >= (1 << int(mjtState.mjNSTATE)):
    raise FatalError('mj_stateSize: invalid state signature %d' % sig)
  c = m._c
  nm = int(getattr(c, 'nmocap', 0))
  sizes = {'TIME': 1, 'QPOS': c.nq, 'QVEL': c.nv, 'ACT': c.na, 'WARMSTART': c.nv, 'CTRL': c.nu, 'QFRC_APPLIED': c.nv,
           'XFRC_APPLIED': 6 * c.nbody, 'EQ_ACTIVE': c.neq, 'MOCAP_POS': 3 * nm, 'MOCAP_QUAT': 4 * nm, 'USERDATA': 0, 'PLUGIN': 0}
  field = {'QPOS': 'qpos', 'QVEL': 'qvel', 'ACT': 'act', 'WARMSTART': 'qacc_warmstart', 'CTRL': 'ctrl', 'QFRC_APPLIED': 'qfrc_applied',
           'XFRC_APPLIED': 'xfrc_applied', 'EQ_ACTIVE': 'eq_active', 'MOCAP_POS': 'mocap_pos', 'MOCAP_QUAT': 'mocap_quat'}
  return [(n, field.get(n), int(sizes[n])) for i, n in enumerate(_STATE_NAMES) if sig & (1 << i)]


def mj_stateSize(m, sig):
  """engine.py:248."""
  return sum(n for _, _, n in _state_parts(m, sig))


def mj_getState(m, d, state, sig):
  """engine.py:249."""
  _check(m, d)
  state = np.asarray(state)
  parts = _state_parts(m, sig)
  if state.size != sum(n for _, _, n in parts):
    raise TypeError('state size should equal mj_stateSize(m, sig)')
  k = 0
  for name, field, n in parts:
    if name == 'TIME':
      state[k] = d.time
    elif field and n:
      state[k:k + n] = np.asarray(d._array(field), dtype=np.float64).ravel()
    k += n


def mj_setState(m, d, state, sig):
  """engine.py:280."""
  _check(m, d)
  state = np.asarray(state, dtype=np.float64).ravel()
  parts = _state_parts(m, sig)
  if state.size != sum(n for _, _, n in parts):
    raise TypeError('state size should equal mj_stateSize(m, sig)')
  k = 0
  for name, field, n in parts:
    if name == 'TIME':
      d.time = state[k]
    elif field and n:
      a = d._array(field)
      a[...] = state[k:k + n].reshape(a.shape)
    k += n


def _bad(a):
  a = np.asarray(a, dtype=np.float64).ravel()
  bad = np.nonzero(~np.isfinite(a) | (np.abs(a) > mjMAXVAL))[0]
  return int(bad[0]) if bad.size else -1


def _host_warning(m, d, which, info):
  n = d._warning.number
  if not int(m._c.opt.disableflags) & C['DMC_DSBL_AUTORESET']:
    keep = n.copy()
    mj_resetData(m, d)
    n[:] = keep
  n[int(which)] += 1
  d._warning.lastinfo[int(which)] = info
  _warn('Nan, Inf or huge value in %s at DOF %d. The simulation is unstable. Time = %.4f.' %
        ({4: 'QPOS', 5: 'QVEL', 6: 'QACC'}[int(which)], info, d.time))


def mj_checkPos(m, d):
  """mj_step's first check, callable on its own (engine_test.py:503-511)."""
  _check(m, d)
  k = _bad(d._array('qpos'))
  if k >= 0:
    _host_warning(m, d, mjtWarning.mjWARN_BADQPOS, k)


def mj_checkVel(m, d):
  _check(m, d)
  k = _bad(d._array('qvel'))
  if k >= 0:
    _host_warning(m, d, mjtWarning.mjWARN_BADQVEL, k)


def mj_checkAcc(m, d):
  _check(m, d)
  k = _bad(d._array('qacc'))
  if k >= 0:
    _host_warning(m, d, mjtWarning.mjWARN_BADQACC, k)


def mj_contactForce(m, d, id_, result):
  """core.py:551: the 6D wrench of contact `id_` in the contact frame (force: normal, tangent, tangent; torque: torsion,
  roll, roll), as of the last mj_forward / mj_fwdConstraint."""
  _check(m, d)
  if not 0 <= int(id_) < d.ncon:
    raise FatalError('mj_contactForce: contact id %d out of range [0, %d)' % (id_, d.ncon))
  w = np.asarray(d._batch.get('contact_force'), dtype=np.float64).reshape(-1, 6)[int(id_)]
  np.asarray(result).reshape(-1)[:6] = w


def mj_objectVelocity(m, d, objtype, objid, res, flg_local):
  """core.py:522: 6D velocity (angular, linear) of a body / xbody / geom / site in the world or the object's own frame."""
  _check(m, d)
  c = m._c
  kinds = {1: ('body', 'xipos', None), 2: ('body', 'xpos', 'xmat'), 5: ('geom', 'geom_xpos', 'geom_xmat'), 6: ('site', 'site_xpos', 'site_xmat')}
  if int(objtype) not in kinds:
    raise FatalError('mj_objectVelocity: invalid object type %d' % int(objtype))
  kind, posf, matf = kinds[int(objtype)]
  get = lambda n, *shape: d._array(n).reshape(shape)
  objid = int(objid)
  body = objid if kind == 'body' else int(c.geom_bodyid[objid]) if kind == 'geom' else int(c.site_bodyid[objid])
  n = {'body': c.nbody, 'geom': c.ngeom, 'site': c.nsite}[kind]
  pos = get(posf, n, 3)[objid]
  if matf is None:
    mat = mjcf_compiler.quat_to_mat(mjcf_compiler.quat_mul(get('xquat', c.nbody, 4)[objid], c.body_iquat[objid]))
  else:
    mat = get(matf, n, 3, 3)[objid]
  cvel = get('cvel', c.nbody, 6)[body]
  com = get('subtree_com', c.nbody, 3)[int(c.body_rootid[body])]
  ang = cvel[:3]
  lin = cvel[3:] - np.cross(pos - com, ang)
  if flg_local:
    ang, lin = mat.T @ ang, mat.T @ lin
  res = np.asarray(res).reshape(-1)
  res[:3], res[3:6] = ang, lin


# ---------------------------------------------------------------------------------------------------------------------
# visualisation structs: plain records (the reference subclasses them at import, wrapper/core.py:573-760); nothing renders
# ---------------------------------------------------------------------------------------------------------------------
class MjvCamera:
  def __init__(self):
    self.type, self.fixedcamid, self.trackbodyid = 0, -1, -1
    self.lookat = np.zeros(3)
    self.distance, self.azimuth, self.elevation = 2.0, 90.0, -45.0


class MjvOption:
  def __init__(self):
    self.label, self.frame = 0, 0
    self.geomgroup = np.array([1, 1, 1, 0, 0, 0], dtype=np.uint8)
    self.sitegroup = np.array([1, 1, 1, 0, 0, 0], dtype=np.uint8)
    self.jointgroup = self.tendongroup = self.actuatorgroup = np.array([1, 1, 1, 0, 0, 0], dtype=np.uint8)
    self.flags = np.zeros(int(mjtVisFlag.mjNVISFLAG), dtype=np.uint8)


class MjvPerturb:
  def __init__(self):
    self.select, self.active = 0, 0
    self.refpos, self.refquat, self.localpos = np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)


class MjvFigure:
  pass


class MjrRect:
  def __init__(self, left=0, bottom=0, width=0, height=0):
    self.left, self.bottom, self.width, self.height = left, bottom, width, height


def _no_render(*args, **kwargs):
  raise NotImplementedError('rendering is not part of the MI355X physics backend (DESIGN.md, out of scope)')


class MjvScene:
  def __init__(self, *args, **kwargs):
    _no_render()


class MjrContext:
  def __init__(self, *args, **kwargs):
    _no_render()


mjv_defaultFreeCamera = mjv_updateScene = mjv_select = mjr_render = mjr_readPixels = mjr_setBuffer = mjr_overlay = _no_render


# ---------------------------------------------------------------------------------------------------------------------
# what dm_control's build generates from MuJoCo's headers (mjbindings/{sizes,enums,constants}.py; absent from the
# reference tree): the same tables for this backend's surface
# ---------------------------------------------------------------------------------------------------------------------
def array_sizes():
  """`mjbindings.sizes.array_sizes`: struct -> field -> (size name | int, ...) for every array MjModel / MjData serve here
  (dm_control/mujoco/index.py:93-174 builds the named indexers from it)."""
  arrays, _ = _model_attr_names()
  probe = mjcf_compiler.compile_xml(
      "<mujoco><worldbody><light name='l'/><camera name='c'/><body name='b'><joint name='j'/><geom name='g' size='.1'/>"
      "<site name='s'/></body></worldbody><actuator><motor name='a' joint='j'/></actuator>"
      "<sensor><jointpos name='p' joint='j'/></sensor></mujoco>")
  prefixes = (('body_', 'nbody'), ('jnt_', 'njnt'), ('dof_', 'nv'), ('geom_', 'ngeom'), ('site_', 'nsite'), ('cam_', 'ncam'),
              ('light_', 'nlight'), ('mat_', 'nmat'), ('actuator_', 'nu'), ('sensor_', 'nsensor'), ('tendon_', 'ntendon'),
              ('wrap_', 'nwrap'), ('eq_', 'neq'), ('key_', 'nkey'), ('pair_', 'npair'), ('numeric_', 'nnumeric'))
  model = {}
  for name in arrays:
    v = getattr(probe, name)
    row = {'qpos0': 'nq', 'qpos_spring': 'nq', 'numeric_data': 'nnumericdata', 'mesh_vert': 'nmeshvert'}.get(name)
    if row is None:
      row = next((s for pre, s in prefixes if name.startswith(pre)), None)
    if row is None or v.ndim not in (1, 2):
      continue
    if name.startswith('key_') and v.ndim == 2:
      cols = {'key_qpos': 'nq', 'key_qvel': 'nv', 'key_act': 'na', 'key_ctrl': 'nu'}.get(name, int(v.shape[1]))
      model[name] = (row, cols)
    else:
      model[name] = (row,) + ((int(v.shape[1]),) if v.ndim == 2 else ())
  model.update({'cam_pos0': ('ncam', 3), 'cam_poscom0': ('ncam', 3), 'cam_mat0': ('ncam', 9), 'light_pos0': ('nlight', 3),
                'light_poscom0': ('nlight', 3), 'light_dir0': ('nlight', 3), 'mesh_normal': ('nmeshnormal', 3),
                'skin_rgba': ('nskin', 4), 'tendon_rgba': ('ntendon', 4), 'actuator_actadr': ('nu',), 'actuator_actnum': ('nu',),
                'M_rownnz': ('nv',), 'M_rowadr': ('nv',), 'M_colind': ('nC',), 'dof_Madr': ('nv',)})
  # (the name_*adr arrays are left to index.py's own rule: 'name_actuatoradr' -> 'nactuator', which nu / na inherit)
  return {'mjmodel': model, 'mjdata': dict(DATA_SHAPES)}


def _namedtuple_enum(e):
  import collections
  members = list(e.__members__.items())
  T = collections.namedtuple(e.__name__, [k for k, _ in members])
  return T(*[int(v) for _, v in members])


def mjbindings_modules():
  """{module name: module} for `dm_control.mujoco.wrapper.mjbindings.{sizes,enums,constants}`."""
  import types
  sizes = types.ModuleType('dm_control.mujoco.wrapper.mjbindings.sizes')
  sizes.array_sizes = array_sizes()
  enums = types.ModuleType('dm_control.mujoco.wrapper.mjbindings.enums')
  for name, val in list(globals().items()):
    if name.startswith('mjt') and isinstance(val, enum.EnumMeta):
      setattr(enums, name, _namedtuple_enum(val))
  constants = types.ModuleType('dm_control.mujoco.wrapper.mjbindings.constants')
  for name, val in list(globals().items()):
    if name.startswith('mj') and isinstance(val, (int, float)) and not isinstance(val, bool):
      setattr(constants, name, val)
  return {sizes.__name__: sizes, enums.__name__: enums, constants.__name__: constants}


def install():
  """Makes `import mujoco` resolve to this module and provides the three generated `mjbindings` modules, so that an
  unmodified dm_control checkout (engine.py, wrapper/core.py, index.py and everything above them) runs on libdmc_hip.so.
  Call before importing dm_control; INTEGRATION.md section 2."""
  import sys
  sys.modules['mujoco'] = sys.modules[__name__]
  sys.modules.update(mjbindings_modules())

