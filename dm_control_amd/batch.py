"""BatchedPhysics: B independent environments stepped by one HIP kernel launch.

Host-side mirror of the reference's `mujoco.Physics` step surface
(dm_control/mujoco/engine.py:139-176,306-343) over the C-ABI in
include/dmc_batch.h.  Arrays are (B, n) on the host; on the device they are
structure-of-arrays (n, B).
"""
import ctypes

import numpy as np

from dm_control_amd import _native
from dm_control_amd import mjcf_compiler

OUT = dict(sensor=1 << 0, xpos=1 << 1, xquat=1 << 2, xmat=1 << 3, xipos=1 << 4,
           geom=1 << 5, site=1 << 6, subtree_com=1 << 7, qacc=1 << 8,
           actuator=1 << 9, contact=1 << 10, qfrc=1 << 11, cvel=1 << 12, contact_ids=1 << 13)
OUT_ALL = 0x7fffffff


class BatchedPhysics:

  def __init__(self, model, batch_size, device_id=0, precision=32, nconmax=0,
               njmax=0, lanes_per_env=0, njcon=0, specialise=None):
    if not isinstance(model, mjcf_compiler.Model):
      raise TypeError('model must be a compiled mjcf_compiler.Model')
    L = _native.lib()
    self.model = model
    self.batch_size = int(batch_size)
    self.precision = precision
    self.device_id = device_id
    ints, reals = model.pack()
    self._model_ptr = ctypes.c_void_p()
    _native.check(L.dmc_model_create(ints.ctypes.data, ints.size, reals.ctypes.data,
                                     reals.size, ctypes.byref(self._model_ptr)))
    self._ptr = ctypes.c_void_p()
    caps = np.array([nconmax, njmax, lanes_per_env, njcon], dtype=np.int32)
    rc = L.dmc_batch_create_caps(self._model_ptr, self.batch_size, device_id, precision,
                                 caps.ctypes.data, caps.size, ctypes.byref(self._ptr))
    if rc != 0:
      L.dmc_model_destroy(self._model_ptr)
      self._model_ptr = None
      _native.check(rc)
    self.legacy_step = True
    # a model without a baked specialised kernel takes the one built for it on demand, if there is one (specialise.py)
    self._user_caps = (int(nconmax), int(njmax), int(njcon))
    from dm_control_amd import specialise as _spec
    self._spec_future = None
    self.specialised = _spec.attach(self, specialise)

  @classmethod
  def from_xml_string(cls, xml_string, batch_size, assets=None, **kw):
    return cls(mjcf_compiler.compile_xml(xml_string, assets), batch_size, **kw)

  def _poll_specialised(self):
    # a specialised kernel that was being compiled in the background takes over at the first launch after its build
    if self._spec_future is not None:
      from dm_control_amd import specialise as _spec
      _spec.poll(self)

  def wait_specialised(self, timeout=None):
    """Blocks until a background build of this model's specialised kernel (specialise.py, the default for models
    without a baked kernel) has finished and switches the batch over.  Callers that record their launches into a HIP
    graph call this before capturing: a graph keeps the kernel it was captured with.  Returns `self.specialised`."""
    if self._spec_future is not None:
      from dm_control_amd import specialise as _spec
      _spec.poll(self, block=True, timeout=timeout)
    return self.specialised

  def set_task_args(self, blob):
    """dmc_batch_set_task_args: the argument block (bytes) of the task epilogue the attached kernel was built with."""
    buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
    _native.check(_native.lib().dmc_batch_set_task_args(self._ptr, ctypes.cast(buf, ctypes.c_void_p), len(blob)))

  def enable_task(self, on):
    """dmc_batch_enable_task: whether the step launches enqueued from now on end with the task epilogue."""
    _native.check(_native.lib().dmc_batch_enable_task(self._ptr, int(bool(on))))

  def close(self):
    L = _native.lib()
    if getattr(self, '_ptr', None):
      L.dmc_batch_destroy(self._ptr)
      self._ptr = None
    if getattr(self, '_model_ptr', None):
      L.dmc_model_destroy(self._model_ptr)
      self._model_ptr = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  # -- info ---------------------------------------------------------------------
  def info(self):
    a = np.zeros(20, dtype=np.int32)
    _native.check(_native.lib().dmc_batch_info(self._ptr, a.ctypes.data))
    keys = ['B', 'precision', 'lanes_per_env', 'waves_per_block', 'envs_per_block',
            'lds_bytes_per_block', 'grid', 'nconmax', 'njmax', 'env_scratch_bytes', 'static_id',
            'jac_kmax', 'table_lds_bytes', 'envs_per_cu', 'njdense', 'njcon', 'stash', 'stash_bytes_per_env',
            'global_scratch_bytes_per_env', 'work_queue']
    return dict(zip(keys, (int(x) for x in a)))

  def _rows(self, name):
    rows, is_int = ctypes.c_int(), ctypes.c_int()
    _native.check(_native.lib().dmc_batch_field_rows(self._ptr, name.encode(),
                                                     ctypes.byref(rows), ctypes.byref(is_int)))
    return rows.value, bool(is_int.value)

  # -- field access -----------------------------------------------------------------
  def get(self, name):
    rows, is_int = self._rows(name)
    if is_int:
      out = np.zeros((self.batch_size, rows), dtype=np.int32)
      _native.check(_native.lib().dmc_batch_get_int(self._ptr, name.encode(), out.ctypes.data))
    else:
      out = np.zeros((self.batch_size, rows), dtype=np.float64)
      if rows:
        _native.check(_native.lib().dmc_batch_get(self._ptr, name.encode(), out.ctypes.data))
    return out

  def set(self, name, value):
    rows, is_int = self._rows(name)
    if not rows:
      return
    dt = np.int32 if is_int else np.float64
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=dt).reshape(
        (-1, rows) if np.ndim(value) > 1 else (1, rows) if np.ndim(value) == 1 else (1, 1)),
        (self.batch_size, rows)))
    if not rows:
      return
    fn = _native.lib().dmc_batch_set_int if is_int else _native.lib().dmc_batch_set
    _native.check(fn(self._ptr, name.encode(), a.ctypes.data))

  # -- asynchronous host transfers (dmc_batch_set_async / get_async / get_wait) ------------------------------------------
  def set_async(self, name, value, stream=None):
    """Writes the (B, rows) host array `value` (float32 or float64; float32 goes onto the wire as it is for an fp32
    batch) to field `name`, ordered on `stream` with the launches.  `value` is consumed before the call returns."""
    rows, is_int = self._rows(name)
    if is_int:
      raise ValueError('%s is an int field' % name)
    if not rows:
      return
    a = np.asarray(value)
    dt = np.float32 if a.dtype == np.float32 else np.float64
    a = np.ascontiguousarray(np.broadcast_to(a.astype(dt, copy=False).reshape((-1, rows) if a.ndim > 1 else (1, rows)), (self.batch_size, rows)))
    _native.check(_native.lib().dmc_batch_set_async(self._ptr, name.encode(), a.ctypes.data, 32 if dt == np.float32 else 64, stream))

  def get_async(self, names, stream=None):
    """Enqueues the read of the real fields `names` as of this point of `stream` (one device-to-host copy for all of
    them); `get_wait` returns the arrays.  One get at a time."""
    names = list(names)
    arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    _native.check(_native.lib().dmc_batch_get_async(self._ptr, len(names), arr, stream))
    self._pending_get = names

  def get_wait(self, dtype=np.float64, copy=True):
    """{name: (B, rows) array of `dtype` (float64 or float32)} of the get enqueued by `get_async`.  copy=False: read-only
    views of the pinned staging in the batch's own precision (no host copy at all), valid until the next read of ANY kind
    from this batch -- `get`, `get_async`, `get_many` all go through the one staging buffer, which a larger read also
    re-allocates: copy what must outlive that."""
    names = getattr(self, '_pending_get', None)
    if names is None:
      raise RuntimeError('get_wait without a pending get_async')
    if not copy:
      L = _native.lib()
      _native.check(L.dmc_batch_get_wait(self._ptr, len(names), (ctypes.c_void_p * len(names))(), 64))
      self._pending_get = None
      out = {}
      for i, n in enumerate(names):
        rows = self._rows(n)[0]
        dt = np.int32 if self._rows(n)[1] else np.float64 if (self.precision == 64 or n == 'time') else np.float32
        if not rows:
          out[n] = np.zeros((self.batch_size, 0), dtype=dt)
          continue
        p = L.dmc_batch_get_staged(self._ptr, i)
        if not p:
          raise _native.NativeError(L.dmc_last_error().decode())
        a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double if dt == np.float64 else ctypes.c_int32 if dt == np.int32 else ctypes.c_float)), shape=(self.batch_size, rows))
        a.flags.writeable = False
        out[n] = a
      return out
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
      raise ValueError('dtype must be float64 or float32')
    outs = [np.empty((self.batch_size, self._rows(n)[0]), dtype=np.int32 if self._rows(n)[1] else dt) for n in names]
    ptrs = (ctypes.c_void_p * len(names))(*[o.ctypes.data if o.size else None for o in outs])
    _native.check(_native.lib().dmc_batch_get_wait(self._ptr, len(names), ptrs, 32 if dt == np.dtype(np.float32) else 64))
    self._pending_get = None
    return dict(zip(names, outs))

  def get_many(self, names, stream=None, dtype=np.float64, copy=True):
    """Several real fields with one device-to-host copy and one wait."""
    self.get_async(names, stream)
    return self.get_wait(dtype, copy=copy)

  def device_ptr(self, name):
    p = _native.lib().dmc_batch_device_ptr(self._ptr, name.encode())
    if not p:
      raise KeyError(name)
    return p

  def bind(self, name, device_ptr):
    _native.check(_native.lib().dmc_batch_bind(self._ptr, name.encode(), device_ptr))

  def set_output_mask(self, mask):
    _native.check(_native.lib().dmc_batch_set_output_mask(self._ptr, int(mask)))

  def set_opt(self, name, value):
    L = _native.lib()
    if isinstance(value, (int, np.integer, bool)) and name in ('disableflags', 'iterations', 'ls_iterations', 'noslip_iterations', 'stash', 'islands'):
      _native.check(L.dmc_batch_set_opt_int(self._ptr, name.encode(), int(value)))
    else:
      _native.check(L.dmc_batch_set_opt_real(self._ptr, name.encode(), float(value)))

  def set_model_real(self, name, values):
    """Rewrites a model constant shared by the whole batch (see dmc_batch_set_model_real)."""
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel())
    _native.check(_native.lib().dmc_batch_set_model_real(self._ptr, name.encode(), a.ctypes.data, a.size))

  # -- per-environment model deltas ---------------------------------------------------------
  def set_env_geoms(self, names):
    """Declares world-fixed geoms whose pose / size differ between environments (dmc_batch_set_env_geoms); their
    values live in the field 'env_geom' ((B, 16 n): pos 3, xmat 9, size 3, rbound 1 per geom)."""
    ids = np.array([self.model.name2id(n, 'geom') for n in names], dtype=np.int32)
    _native.check(_native.lib().dmc_batch_set_env_geoms(self._ptr, ids.size, ids.ctypes.data))
    self.env_geoms = list(names)

  def pack_env_geom(self, name, pos, quat, size):
    """(B, 16) rows of one declared geom from per-env pos (B, 3) / quat (B, 4) / size (B, 3)."""
    g = self.model.name2id(name, 'geom')
    pos, quat, size = (np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (self.batch_size, n)))
                       for a, n in ((pos, 3), (quat, 4), (size, 3)))
    out = np.zeros((self.batch_size, 16))
    for e in range(self.batch_size):
      _native.check(_native.lib().dmc_env_geom_pack(int(self.model.geom_type[g]), pos[e].ctypes.data, quat[e].ctypes.data,
                                                    size[e].ctypes.data, out[e].ctypes.data))
    return out

  def set_env_geom(self, name, pos=None, quat=None, size=None):
    """Writes one declared geom's per-env values; arguments left None keep the model's."""
    g, k = self.model.name2id(name, 'geom'), self.env_geoms.index(name)
    m = self.model
    rows = self.pack_env_geom(name, m.geom_pos[g] if pos is None else pos, m.geom_quat[g] if quat is None else quat,
                              m.geom_size[g] if size is None else size)
    cur = self.get('env_geom')
    cur[:, 16*k:16*k + 16] = rows
    self.set('env_geom', cur)

  # -- the hot path ---------------------------------------------------------------------
  def set_control(self, control):
    self.set('ctrl', control)

  def step(self, nstep=1, stream=None, forward_after=False):
    """Physics.step(nstep) for every environment in one launch.  `forward_after` (legacy steps): the launch ends with
    the rest of mj_forward at the new state -- dmc_batch_step's legacy_step 2."""
    self._poll_specialised()
    legacy = int(self.legacy_step)
    if forward_after:
      if not legacy:
        raise ValueError('forward_after needs legacy_step')
      legacy = 2
    _native.check(_native.lib().dmc_batch_step(self._ptr, int(nstep), legacy, stream))

  def set_step_probe(self, geom_id, out_ptr, capacity):
    """dmc_batch_set_step_probe: geom `geom_id`'s world position after every physics step of a step launch goes to the
    caller's (capacity, 3, B) device array at `out_ptr` (None: off)."""
    _native.check(_native.lib().dmc_batch_set_step_probe(self._ptr, int(geom_id), ctypes.c_void_p(out_ptr) if out_ptr else None, int(capacity)))

  def rollout(self, nsteps, n_sub_steps=1, ctrl_seq=None, qpos_seq=None, qvel_seq=None, sensordata_seq=None,
              stream=None):
    """`nsteps` env-steps in ONE launch; *_seq are device pointers to (nsteps, rows, B)
    arrays in batch precision (None: skip)."""
    self._poll_specialised()
    _native.check(_native.lib().dmc_batch_rollout(self._ptr, int(nsteps), int(n_sub_steps), ctrl_seq, qpos_seq,
                                                  qvel_seq, sensordata_seq, stream))

  def step1(self, stream=None):
    """mujoco.mj_step1 on its own (engine.py:160): position / velocity stage, state unchanged."""
    _native.check(_native.lib().dmc_batch_step1(self._ptr, stream))

  def step2(self, stream=None):
    """mujoco.mj_step2 on its own (engine.py:156): acceleration stage + Euler integration."""
    _native.check(_native.lib().dmc_batch_step2(self._ptr, stream))

  def forward(self, disable_actuation=False, stream=None):
    self._poll_specialised()
    _native.check(_native.lib().dmc_batch_forward(self._ptr, int(disable_actuation), stream))

  def reset(self, env_mask=None, keyframe_id=None):
    m = None
    if env_mask is not None:
      m = np.ascontiguousarray(np.asarray(env_mask, dtype=np.uint8))
    _native.check(_native.lib().dmc_batch_reset(
        self._ptr, m.ctypes.data if m is not None else None,
        -1 if keyframe_id is None else int(keyframe_id)))

  def sync(self):
    _native.check(_native.lib().dmc_batch_sync(self._ptr))

  def invalidate(self, stream=None):
    """Call after writing qpos / qvel / act through memory bound with `bind` (see dmc_batch_invalidate).  With a HIP
    stream handle the invalidation is ordered on that stream (capturable into a HIP graph) instead of synchronous."""
    if stream is None:
      _native.check(_native.lib().dmc_batch_invalidate(self._ptr))
    else:
      _native.check(_native.lib().dmc_batch_invalidate_async(self._ptr, ctypes.c_void_p(stream)))

  RAND_LIMITED, RAND_UNLIMITED_HINGE, RAND_QUATERNION, RAND_FREE_NORMAL, RAND_ALL = 1, 2, 4, 8, 7

  def randomize_joints(self, seed, draw_ptr, env_mask_ptr=None, flags=7, stream=None):
    """randomize_limited_and_rotational_joints (suite/utils/randomizers.py:35-88) on the device for the environments
    whose entry of the (B,) int32 device array at `env_mask_ptr` is non-zero (None: all); `draw_ptr`: (B,) int32 device
    array of per-environment draw counters, incremented for the environments drawn (dmc_batch_randomize_joints)."""
    _native.check(_native.lib().dmc_batch_randomize_joints(self._ptr, ctypes.c_uint64(int(seed) & (2**64 - 1)),
                                                           ctypes.c_void_p(draw_ptr), ctypes.c_void_p(env_mask_ptr), int(flags),
                                                           ctypes.c_void_p(stream) if stream else None))

  def enable_profiling(self, enabled=True):
    """Brackets every launch with hipEvents (dmc_batch_enable_profiling; engine.py:135-137 enable_profiling)."""
    _native.check(_native.lib().dmc_batch_enable_profiling(self._ptr, int(bool(enabled))))

  def timer(self, which=0):
    """(seconds, count) of timer 0 = mjTIMER_STEP / 1 = mjTIMER_FORWARD; waits for the launches issued so far."""
    dur, num = ctypes.c_double(), ctypes.c_longlong()
    _native.check(_native.lib().dmc_batch_get_timer(self._ptr, int(which), ctypes.byref(dur), ctypes.byref(num)))
    return dur.value, num.value

  def time_steps(self, nstep, reps, stream=None):
    ms = ctypes.c_float()
    _native.check(_native.lib().dmc_batch_time_steps(self._ptr, int(nstep), int(self.legacy_step),
                                                     int(reps), stream, ctypes.byref(ms)))
    return ms.value

  PROF_NAMES = ['load', 'kinematics', 'com_pos', 'crb_chol', 'collision', 'constraint', 'com_vel', 'rne',
                'sensors', 'actuation', 'fwd_acc', 'sol_init', 'sol_grad', 'sol_linesearch', 'sol_update',
                'euler', 'trailing_step1', 'store', 'noslip', 'sol_hess', 'sol_factor', 'sol_solve', 'ls_setup',
                # the sub-markers (step_core.h PROF_X1..X8) by where they stand: composite inertias / entries of M (mj_crb), the
                # factorisations of M and M + h D (mj_factorM: position stage and Euler), the joints' local poses / the
                # composition along the chains (mj_kinematics), noslip's A = J_F M^-1 J_F' / its sweeps, and the tail of
                # fwd_constraint after the solver (forces at the solution, qfrc_constraint = J' f)
                'crb_composite', 'crb_entries', 'factor_M', 'kin_local_poses', 'kin_compose', 'noslip_build_A',
                'noslip_sweeps', 'constraint_tail']

  def prof_enable(self, on=True):
    _native.check(_native.lib().dmc_batch_prof_enable(self._ptr, int(on)))

  def prof_get(self):
    buf = np.zeros(32)
    n = ctypes.c_int()
    _native.check(_native.lib().dmc_batch_prof_get(self._ptr, buf.ctypes.data, ctypes.byref(n)))
    return dict(zip(self.PROF_NAMES, buf[:len(self.PROF_NAMES)]))

  # -- debug ----------------------------------------------------------------------------
  def wave_trace(self, enable=None):
    """enable=True/False switches the trace; no argument: (8, 8, nitems) int array, a ring of the last 8 launches
    (slot = launch % 8 since enabling): kernel entry, start and end of every wave item on the 100 MHz constant clock,
    workgroup index, then the clock after the opening position / velocity stage, the first acceleration stage, the
    first integration and the trailing stage."""
    L = _native.lib()
    n = ctypes.c_int(0)
    if enable is not None:
      _native.check(L.dmc_batch_wave_trace(self._ptr, int(bool(enable)), None, ctypes.byref(n)))
      return None
    info = self.info()
    out = np.zeros((8, 8, (info['B'] * info['lanes_per_env'] + 63) // 64), dtype=np.int32)
    _native.check(L.dmc_batch_wave_trace(self._ptr, 1, out.ctypes.data, ctypes.byref(n)))
    return out

  def debug_enable(self, n):
    _native.check(_native.lib().dmc_batch_debug_enable(self._ptr, int(n)))

  def debug_get(self, name, env=0, maxcount=1 << 16):
    buf = np.zeros(maxcount)
    cnt = ctypes.c_int()
    _native.check(_native.lib().dmc_batch_debug_get(self._ptr, name.encode(), env, buf.ctypes.data, ctypes.byref(cnt)))
    return buf[:cnt.value].copy()
