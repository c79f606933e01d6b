"""ctypes wrapper around oracle/libmjstep_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It mirrors the slice of the reference's `Physics` facade that the
parity tests need (dm_control/mujoco/engine.py:139-176,306-343): set_control,
step (legacy / non-legacy), forward, reset, and named access to mjData fields.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libmjstep_oracle.so')
_lib = None


def build(force=False):
  src = os.path.join(_HERE, 'mjstep_oracle.c')
  hdr = os.path.join(os.path.dirname(_HERE), 'include', 'dmc_model_layout.h')
  stale = (not os.path.exists(_LIB_PATH) or
           os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
  if force or stale:
    subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
  return _LIB_PATH


def lib():
  global _lib
  if _lib is None:
    build()
    L = ctypes.CDLL(_LIB_PATH)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    pi, pd = ctypes.POINTER(ci), ctypes.POINTER(cd)
    L.ora_last_error.restype = ctypes.c_char_p
    L.ora_model_create.restype = vp
    L.ora_model_create.argtypes = [ctypes.POINTER(ctypes.c_int32), ci, pd, ci]
    L.ora_model_free.argtypes = [vp]
    L.ora_model_opt_int.restype = ci
    L.ora_model_opt_int.argtypes = [vp, ctypes.c_char_p, ci, ci]
    L.ora_model_opt_real.restype = cd
    L.ora_model_opt_real.argtypes = [vp, ctypes.c_char_p, ci, cd]
    L.ora_model_int_field.restype = pi
    L.ora_model_int_field.argtypes = [vp, ctypes.c_char_p, pi]
    L.ora_model_real_field.restype = pd
    L.ora_model_real_field.argtypes = [vp, ctypes.c_char_p, pi]
    L.ora_data_create.restype = vp
    L.ora_data_create.argtypes = [vp]
    L.ora_data_free.argtypes = [vp]
    L.ora_data_copy.argtypes = [vp, vp, vp]
    L.ora_data_field.restype = pd
    L.ora_data_field.argtypes = [vp, vp, ctypes.c_char_p, pi]
    L.ora_data_int.restype = ci
    L.ora_data_int.argtypes = [vp, ctypes.c_char_p]
    L.ora_data_warning.restype = pi
    L.ora_data_warning.argtypes = [vp]
    L.ora_data_efc_int.restype = pi
    L.ora_data_efc_int.argtypes = [vp, ctypes.c_char_p]
    L.ora_data_contact.argtypes = [vp, ci, pd]
    L.ora_reset.argtypes = [vp, vp, ci]
    for f in ('ora_forward', 'ora_step1', 'ora_step2'):
      getattr(L, f).argtypes = [vp, vp]
    L.ora_step.argtypes = [vp, vp, ci]
    L.ora_physics_step_legacy.argtypes = [vp, vp, ci]
    L.ora_physics_step_legacy_many.argtypes = [vp, ctypes.POINTER(vp), ci, ci]
    L.ora_rollout_legacy.argtypes = [vp, ctypes.POINTER(vp), ci, ci, ci, pd]
    L.ora_contact_force.argtypes = [vp, vp, ci, pd]
    L.ora_object_velocity.argtypes = [vp, vp, ci, ci, ci, pd]
    _lib = L
  return _lib


class OracleModel:

  def __init__(self, compiled):
    self.compiled = compiled
    ints, reals = compiled.pack()
    self._ints, self._reals = ints, reals
    self.ptr = lib().ora_model_create(
        ints.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ints.size,
        reals.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), reals.size)
    if not self.ptr:
      raise ValueError(lib().ora_last_error().decode())

  def __del__(self):
    if getattr(self, 'ptr', None):
      lib().ora_model_free(self.ptr)
      self.ptr = None

  def opt_int(self, name, value=None):
    return lib().ora_model_opt_int(self.ptr, name.encode(), value is not None, value or 0)

  def opt_real(self, name, value=None):
    return lib().ora_model_opt_real(self.ptr, name.encode(), value is not None,
                                    0.0 if value is None else value)

  def field(self, name):
    """Writable numpy view of a model array (tests mutate e.g. geom_condim)."""
    n = ctypes.c_int(0)
    p = lib().ora_model_real_field(self.ptr, name.encode(), ctypes.byref(n))
    if not p:
      p = lib().ora_model_int_field(self.ptr, name.encode(), ctypes.byref(n))
      if not p:
        raise KeyError(name)
    return np.ctypeslib.as_array(p, shape=(n.value,)) if n.value else np.zeros(0)


class OraclePhysics:
  """One fp64 CPU environment."""

  def __init__(self, model, legacy_step=True):
    self.model = model if isinstance(model, OracleModel) else OracleModel(model)
    self.m = self.model.compiled
    self.legacy_step = legacy_step
    self.ptr = lib().ora_data_create(self.model.ptr)

  def __del__(self):
    if getattr(self, 'ptr', None):
      lib().ora_data_free(self.ptr)
      self.ptr = None

  def field(self, name):
    n = ctypes.c_int(0)
    p = lib().ora_data_field(self.model.ptr, self.ptr, name.encode(), ctypes.byref(n))
    if not p:
      raise KeyError(name)
    if n.value == 0:
      return np.zeros(0)
    return np.ctypeslib.as_array(p, shape=(n.value,))

  def __getattr__(self, name):
    if name.startswith('_') or name in ('model', 'm', 'ptr', 'legacy_step'):
      raise AttributeError(name)
    if name in ('ncon', 'nefc', 'solver_iter', 'nisland'):
      return lib().ora_data_int(self.ptr, name.encode())
    if name in ('efc_type', 'efc_id', 'efc_state'):
      p = lib().ora_data_efc_int(self.ptr, name.encode())
      return np.ctypeslib.as_array(p, shape=(max(self.nefc, 1),))[:self.nefc]
    try:
      return self.field(name)
    except KeyError:
      raise AttributeError(name)

  @property
  def time(self):
    return float(self.field('time')[0])

  @time.setter
  def time(self, v):
    self.field('time')[0] = v

  def efc_island(self):
    """mjData.efc_island of the last solve (island of every constraint row, -1: none)."""
    return np.array(np.ctypeslib.as_array(lib().ora_data_efc_int(self.ptr, b'efc_island'), shape=(max(self.nefc, 1),))[:self.nefc])

  def dof_island(self):
    """mjData.dof_island of the last solve (island of every dof, -1: its tree has no constraint)."""
    return np.array(np.ctypeslib.as_array(lib().ora_data_efc_int(self.ptr, b'dof_island'), shape=(len(self.qvel),)))

  @property
  def warning(self):
    return np.ctypeslib.as_array(lib().ora_data_warning(self.ptr), shape=(9,))

  def contact(self, i):
    out = np.zeros(30)
    lib().ora_data_contact(self.ptr, i, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return dict(dist=out[0], pos=out[1:4], frame=out[4:13].reshape(3, 3),
                includemargin=out[13], friction=out[14:19], solref=out[19:21],
                solimp=out[21:26], dim=int(out[26]), geom1=int(out[27]),
                geom2=int(out[28]), efc_address=int(out[29]))

  def contact_force(self, i):
    out = np.zeros(6)
    lib().ora_contact_force(self.model.ptr, self.ptr, i,
                            out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out.reshape(2, 3)

  def object_velocity(self, objtype, objid, local=False):
    out = np.zeros(6)
    lib().ora_object_velocity(self.model.ptr, self.ptr, objtype, objid, int(local),
                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out.reshape(2, 3)

  def set_control(self, ctrl):
    np.copyto(self.ctrl, ctrl)

  def forward(self):
    lib().ora_forward(self.model.ptr, self.ptr)

  def step1(self):
    lib().ora_step1(self.model.ptr, self.ptr)

  def step2(self):
    lib().ora_step2(self.model.ptr, self.ptr)

  def mj_step(self, nstep=1):
    lib().ora_step(self.model.ptr, self.ptr, nstep)

  def step(self, nstep=1):
    if self.legacy_step:
      lib().ora_physics_step_legacy(self.model.ptr, self.ptr, nstep)
    else:
      lib().ora_step(self.model.ptr, self.ptr, nstep)

  def reset(self, keyframe_id=None):
    lib().ora_reset(self.model.ptr, self.ptr, -1 if keyframe_id is None else keyframe_id)
    # engine.py:326-327: forward with actuation disabled
    flags = self.model.opt_int('disableflags')
    self.model.opt_int('disableflags', flags | (1 << 11))
    self.forward()
    self.model.opt_int('disableflags', flags)

  def after_reset(self):
    flags = self.model.opt_int('disableflags')
    self.model.opt_int('disableflags', flags | (1 << 11))
    self.forward()
    self.model.opt_int('disableflags', flags)

  def copy(self):
    other = OraclePhysics(self.model, self.legacy_step)
    lib().ora_data_copy(self.model.ptr, other.ptr, self.ptr)
    return other


def rollout_legacy(physics_list, actions, nsub=1):
  """Steps every OraclePhysics in `physics_list` through actions (T, B, nu) in C
  (single thread).  Used by the CPU baseline and by parity tests."""
  B = len(physics_list)
  actions = np.ascontiguousarray(actions, dtype=np.float64)
  T = actions.shape[0]
  assert actions.shape[1] == B
  arr = (ctypes.c_void_p * B)(*[p.ptr for p in physics_list])
  lib().ora_rollout_legacy(physics_list[0].model.ptr, arr, B, T, nsub,
                           actions.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
