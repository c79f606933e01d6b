"""`Physics`: the reference's `mujoco.Physics` facade over the HIP batch.

Mirrors dm_control/mujoco/engine.py:83-622 for the step surface: from_xml_string
/ from_xml_path (:446-476), step (:164), forward (:335), reset (:306),
after_reset (:329), reset_context, set_control (:139), get_state / set_state
(:235-285), copy (:287), check_invalid_state (:345-368), suppress_physics_errors
(:125), control / position / velocity / activation / state / time / timestep
(:589-622), `.model`, `.data`, `.named.{model,data}`, and `action_spec` (:1093).
Rendering (Camera, render) is out of scope (no GL on the compute path).

batch_size == 1 (default): arrays have the reference's shapes, so suite tasks are
drop-in.  batch_size == B > 1: every data array gains a leading batch dimension.

`physics.data` is a host mirror of the device arrays: reads fetch lazily (and
are cached until the next step/forward/reset), writes to the input fields
(`qpos qvel act ctrl qacc_warmstart qfrc_applied xfrc_applied mocap_pos mocap_quat time`) are uploaded before the
next kernel launch.  High-throughput callers use `physics.batch`
(`BatchedPhysics`: device pointers, zero-copy binds) instead of the mirror.
"""
import contextlib
import logging

import numpy as np

from dm_control_amd import mjcf_compiler
from dm_control_amd.batch import BatchedPhysics
from dm_control_amd.envs import control
from dm_control_amd.envs.dm_env_api import specs

mjMAXVAL = mjcf_compiler.C['DMC_MAXVAL']
_WARNING_NAMES = ['mjWARN_INERTIA', 'mjWARN_CONTACTFULL', 'mjWARN_CNSTRFULL', 'mjWARN_VGEOMFULL',
                  'mjWARN_BADQPOS', 'mjWARN_BADQVEL', 'mjWARN_BADQACC', 'mjWARN_BADCTRL',
                  'dmcWARN_COLLISION']
# model arrays tasks may rewrite through physics.model / physics.named.model between
# episodes; changes are pushed to the device tables before the next launch
_MUTABLE_MODEL_FIELDS = ('dof_damping', 'jnt_stiffness', 'jnt_range', 'jnt_margin', 'qpos_spring', 'site_pos',
                         'site_quat', 'site_size', 'actuator_ctrlrange', 'actuator_forcerange', 'wrap_prm', 'body_pos',
                         'body_quat',
                         # geom frames / sizes as tasks rewrite them (suite/reacher.py:88-94, suite/fish.py:150-154:
                         # the target geom); like MuJoCo, nothing derived at compile time (inertias, geom_rbound) follows
                         'geom_pos', 'geom_quat', 'geom_size')
# rendering attributes: writable host arrays (suite/finger.py:139-140 site_rgba, suite/fish.py:115 geom_rgba,
# suite/swimmer.py light_pos, suite/base.py:104-112 mat_rgba); they never reach the device
_HOST_ONLY_MODEL_FIELDS = ('geom_rgba', 'site_rgba', 'mat_rgba', 'light_pos', 'light_dir',
                           'body_sameframe', 'body_simple', 'geom_sameframe', 'site_sameframe')
# the collision filter bits: a write changes WHICH geom pairs the device tests -- a table that is laid out at batch creation
# -- so the batch is rebuilt from the edited model before the next launch (composer/initializers/prop_initializer.py:138-160
# switches props' contacts off while it places them, then back on); see Physics._rebuild_batch
_STRUCTURAL_MODEL_FIELDS = ('geom_contype', 'geom_conaffinity')
_INVALID_PHYSICS_STATE = ('Physics state is invalid. Warning(s) raised: {warning_names}')

_INPUT_FIELDS = ('qpos', 'qvel', 'act', 'ctrl', 'qacc_warmstart', 'qfrc_applied', 'xfrc_applied', 'time', 'mocap_pos',
                 'mocap_quat')
_INT_FIELDS = ('ncon', 'nefc', 'solver_iter', 'warning', 'contact_geom1', 'contact_geom2')
# field -> (row object kind for named access, columns per row)
_FIELD_AXES = {
    'qpos': ('joint_q', None), 'qvel': ('joint_v', None), 'qacc': ('joint_v', None),
    'qacc_warmstart': ('joint_v', None), 'qfrc_applied': ('joint_v', None), 'xfrc_applied': ('body', 6),
    'mocap_pos': ('mocap', 3), 'mocap_quat': ('mocap', 4),      # rows named after the mocap bodies (mujoco/index.py:177-267)
    'qfrc_actuator': ('joint_v', None), 'qfrc_bias': ('joint_v', None),
    'qfrc_constraint': ('joint_v', None),
    'ctrl': ('actuator', None), 'actuator_force': ('actuator', None),
    'act': ('act', None),      # rows named after the actuators that have an activation state (index.py:93-99 'na')
    'sensordata': ('sensor', None),
    'xpos': ('body', 3), 'xquat': ('body', 4), 'xmat': ('body', 9), 'xipos': ('body', 3),
    'subtree_com': ('body', 3), 'geom_xpos': ('geom', 3), 'geom_xmat': ('geom', 9),
    'site_xpos': ('site', 3), 'site_xmat': ('site', 9),
    'xanchor': ('joint', 3), 'xaxis': ('joint', 3),      # derived on the host (_Data._joint_frames)
    'ten_length': ('tendon', None), 'ten_velocity': ('tendon', None),      # derived on the host (_Data._tendons)
    'cvel': ('body', 6),      # com-based body velocities (rotational, translational): soccer/observables.py:278 reads them
}
_COLS = {3: ['x', 'y', 'z'], 4: ['qw', 'qx', 'qy', 'qz'], 6: ['fx', 'fy', 'fz', 'tx', 'ty', 'tz'],
         9: ['xx', 'xy', 'xz', 'yx', 'yy', 'yz', 'zx', 'zy', 'zz']}


_NTIMER = 15      # mjNTIMER


class _Timer:
  """One mjData.timer[k] record, read from the device batch at access time."""

  def __init__(self, physics, which):
    self._p, self._k = physics, which

  def _read(self):
    return self._p.batch.timer(self._k) if self._k < 2 else (0.0, 0)

  @property
  def duration(self):
    return self._read()[0]

  @property
  def number(self):
    return self._read()[1]


class _Data:
  """Host mirror of the batched mjData arrays (see module docstring)."""

  def __init__(self, physics):
    object.__setattr__(self, '_p', physics)
    object.__setattr__(self, '_cache', {})
    object.__setattr__(self, '_touched', set())
    object.__setattr__(self, '_shadow', {})      # view semantics: what the device holds of each handed-out input array
    object.__setattr__(self, '_written', set())  # inputs assigned through data.<name> = ... since the last launch
    object.__setattr__(self, '_reads', set())    # device fields fetched since the last launch ...
    object.__setattr__(self, '_habit', ())       # ... and those fetched after the launch before: prefetched together
    object.__setattr__(self, '_prefetched', {})

  @property
  def ptr(self):
    """MjData.ptr: an opaque handle (see Model.ptr)."""
    return self

  def _joint_frames(self):
    """mjData.xanchor / xaxis (joint anchors and axes in the world frame): mj_kinematics' joint loop replayed on the
    host from qpos and the parents' frames -- each joint's anchor and axis are taken in the body frame accumulated
    BEFORE that joint moves it, so bodies with several joints cannot be served from the final xpos / xmat."""
    p, m = self._p, self._p.model
    B = p.batch_size
    qpos = np.asarray(p.batch.get('qpos'), dtype=np.float64).reshape(B, -1)
    xpos = np.asarray(p.batch.get('xpos'), dtype=np.float64).reshape(B, -1, 3)
    xquat = np.asarray(p.batch.get('xquat'), dtype=np.float64).reshape(B, -1, 4)
    mpos = np.asarray(p.batch.get('mocap_pos'), dtype=np.float64).reshape(B, -1, 3) if getattr(m, 'nmocap', 0) else None
    mquat = np.asarray(p.batch.get('mocap_quat'), dtype=np.float64).reshape(B, -1, 4) if getattr(m, 'nmocap', 0) else None
    C = mjcf_compiler
    anchor, axis = np.zeros((B, m.njnt, 3)), np.zeros((B, m.njnt, 3))
    for e in range(B):
      for b in range(1, m.nbody):
        j0, jn = int(m.body_jntadr[b]), int(m.body_jntnum[b])
        if jn == 0:
          continue
        if jn == 1 and m.jnt_type[j0] == 0:      # free joint
          qa = int(m.jnt_qposadr[j0])
          anchor[e, j0] = qpos[e, qa:qa + 3]
          axis[e, j0] = m.jnt_axis[j0]
          continue
        pid = int(m.body_parentid[b])
        bp, bq = m.body_pos[b], m.body_quat[b]
        if getattr(m, 'nmocap', 0) and m.body_mocapid[b] >= 0:
          bp, bq = mpos[e, m.body_mocapid[b]], mquat[e, m.body_mocapid[b]] / np.linalg.norm(mquat[e, m.body_mocapid[b]])
        pos = xpos[e, pid] + C.quat_to_mat(xquat[e, pid]) @ bp if pid else np.array(bp, dtype=np.float64)
        quat = C.quat_mul(xquat[e, pid], bq) if pid else np.array(bq, dtype=np.float64)
        for j in range(j0, j0 + jn):
          R = C.quat_to_mat(quat)
          axis[e, j] = R @ m.jnt_axis[j]
          anchor[e, j] = R @ m.jnt_pos[j] + pos
          qa, t = int(m.jnt_qposadr[j]), int(m.jnt_type[j])
          if t == 2:      # slide
            pos = pos + axis[e, j] * (qpos[e, qa] - m.qpos0[qa])
          else:           # ball / hinge: rotate about the anchor
            if t == 1:
              qloc = qpos[e, qa:qa + 4] / np.linalg.norm(qpos[e, qa:qa + 4])
            else:
              qloc = C.axisangle_to_quat(m.jnt_axis[j], qpos[e, qa] - m.qpos0[qa])
            quat = C.quat_mul(quat, qloc)
            pos = anchor[e, j] - C.quat_to_mat(quat) @ m.jnt_pos[j]
    return anchor, axis

  def _tendons(self):
    """mjData.ten_length / ten_velocity (mj_tendon, mj_fwdVelocity: `ten_velocity = ten_J qvel`) from the device's state
    as of the last launch -- the kernel re-derives the few tendon lengths where it needs them and stores none.  Fixed
    tendons: the coefficient-weighted sum of joint coordinates / velocities; site-to-site spatial tendons: the segment
    lengths, and their rates from the sites' velocities (com-based `cvel` of the body, moved to the site)."""
    p, m = self._p, self._p.model
    B, nt = p.batch_size, m.ntendon
    length, velocity = np.zeros((B, nt)), np.zeros((B, nt))
    get = lambda n, *shape: np.asarray(p.batch.get(n), dtype=np.float64).reshape((B,) + shape)
    qpos, qvel = get('qpos', m.nq), get('qvel', m.nv)
    spatial = [t for t in range(nt) if m.tendon_num[t] and m.wrap_type[m.tendon_adr[t]] != mjcf_compiler.C['DMC_WRAP_JOINT']]
    if spatial:
      sx, cvel, com = get('site_xpos', m.nsite, 3), get('cvel', m.nbody, 6), get('subtree_com', m.nbody, 3)
    for t in range(nt):
      w0, wn = int(m.tendon_adr[t]), int(m.tendon_num[t])
      if t not in spatial:
        for w in range(w0, w0 + wn):
          j = int(m.wrap_objid[w])
          length[:, t] += m.wrap_prm[w] * qpos[:, m.jnt_qposadr[j]]
          velocity[:, t] += m.wrap_prm[w] * qvel[:, m.jnt_dofadr[j]]
        continue
      def point(w):
        sid = int(m.wrap_objid[w])
        b = int(m.site_bodyid[sid])
        pos = sx[:, sid]
        return pos, cvel[:, b, 3:] + np.cross(cvel[:, b, :3], pos - com[:, m.body_rootid[b]])
      for w in range(w0, w0 + wn - 1):
        (p0, v0), (p1, v1) = point(w), point(w + 1)
        dif = p1 - p0
        n = np.linalg.norm(dif, axis=-1)
        length[:, t] += n
        ok = n > mjcf_compiler.C['DMC_MINVAL']
        velocity[ok, t] += np.einsum('ek,ek->e', dif[ok] / n[ok, None], (v1 - v0)[ok])
    return length, velocity

  def _fetch(self, name):
    p = self._p
    if name in ('ten_length', 'ten_velocity'):
      a = self._tendons()[name == 'ten_velocity']
      return a[0] if p.batch_size == 1 else a
    if name in ('xanchor', 'xaxis'):
      # (the device's state: like every derived array, as of the last launch -- edits not yet forwarded are not seen)
      anchor, axis = self._joint_frames()
      a = anchor if name == 'xanchor' else axis
      return a[0] if p.batch_size == 1 else a
    a = self._prefetched.pop(name, None)
    if a is None:
      a = self._fetch_device(name)
    else:
      self._reads.add(name)      # (served from the prefetch: still part of what this loop reads)
    rows = a.shape[1]
    if name in _FIELD_AXES and _FIELD_AXES[name][1]:
      c = _FIELD_AXES[name][1]
      a = a.reshape(a.shape[0], rows // c, c)
    if name in ('time', 'ncon', 'nefc', 'solver_iter'):
      a = a[:, 0]
    return a[0] if p.batch_size == 1 else a

  def _fetch_device(self, name):
    """One field from the device.  A host loop reads the same few fields after every step (a suite task: qpos, qvel,
    sensordata, ...): the real fields read after the PREVIOUS launch are fetched together with this one -- one
    device-to-host copy and one wait (dmc_batch_get_async / get_wait) instead of one synchronous round trip per field."""
    b = self._p.batch
    self._reads.add(name)
    want = [n for n in self._habit if n != name and n not in self._cache and n not in self._prefetched]
    if want and hasattr(b, 'get_many'):
      want = [n for n in want if n not in ('xanchor', 'xaxis', 'contact', 'ten_length', 'ten_velocity')][:7]      # (a get holds at most 8 fields)
      if want:
        try:
          got = b.get_many([name] + want)
          for n in want:
            self._prefetched[n] = got[n]
          return got[name]
        except Exception:      # pylint: disable=broad-except
          pass                 # (a field the batch does not have: fall back to the plain read)
    return b.get(name)

  @property
  def timer(self):
    """mjData.timer: mjNTIMER records with `.duration` (seconds) and `.number`; [0] = mjTIMER_STEP, [1] =
    mjTIMER_FORWARD are live once `Physics.enable_profiling()` was called (suite/wrappers/mujoco_profiling.py:94-103
    reads `timer[0]`), the others stay zero: the fused launch has no host-visible sub-stages."""
    return [_Timer(self._p, k) for k in range(_NTIMER)]

  def __getattr__(self, name):
    if name.startswith('_'):
      raise AttributeError(name)
    if name == 'time' and self._p.batch_size == 1:
      return float(self._get('time'))
    return self._get(name)

  def _get(self, name):
    if name not in self._cache:
      try:
        self._cache[name] = self._fetch(name)
      except Exception as e:
        raise AttributeError('%s (%s)' % (name, e))
    if name in _INPUT_FIELDS:
      if self._p.view_semantics:
        a = self._cache[name]
        if name not in self._shadow and isinstance(a, np.ndarray) and a.ndim:
          self._shadow[name] = np.array(a, copy=True)      # writes are found by comparison at upload time
      else:
        # a read hands out an array the caller may write into: kept beside a copy, and uploaded only if it differs
        self._touched.add(name)
        a = self._cache[name]
        if name not in self._shadow and isinstance(a, np.ndarray) and a.ndim:
          self._shadow[name] = np.array(a, copy=True)
    return self._cache[name]

  def __setattr__(self, name, value):
    if name not in _INPUT_FIELDS:
      raise AttributeError('data.%s is read-only' % name)
    if name not in self._cache and name != 'time' and np.ndim(value) >= 1:
      # a whole-array assignment (set_control's `data.ctrl = action`): nothing to fetch from the device first
      m = self._p.model
      n = dict(qpos=m.nq, qvel=m.nv, act=m.na, ctrl=m.nu, qacc_warmstart=m.nv, qfrc_applied=m.nv, xfrc_applied=6 * m.nbody,
               mocap_pos=3 * getattr(m, 'nmocap', 0), mocap_quat=4 * getattr(m, 'nmocap', 0))[name]
      c = _FIELD_AXES[name][1]
      shape = (n // c, c) if c else (n,)
      if self._p.batch_size > 1:
        shape = (self._p.batch_size,) + shape
      if np.shape(value) == shape or (np.size(value) == int(np.prod(shape)) and not self._p.view_semantics):
        self._cache[name] = np.empty(shape, dtype=np.float64)
        self._shadow.pop(name, None)
    cur = self._get(name)
    if np.ndim(cur) == 0:
      self._cache[name] = np.asarray(float(value))
    else:
      cur[...] = value
    self._touched.add(name)
    self._written.add(name)

  def _upload(self):
    p = self._p
    p._push_model()
    if p.view_semantics:
      # handed-out input arrays may have been written through any reference to them (mjcf bindings keep the ndarray
      # itself): whatever differs from what the device holds goes up
      for name, dev in self._shadow.items():
        cur = self._cache.get(name)
        if isinstance(cur, np.ndarray) and cur.ndim and not np.array_equal(cur, dev, equal_nan=True):
          self._touched.add(name)
    send = getattr(p.batch, 'set_async', None) or p.batch.set      # (stream-ordered with the launches: the facade uses the null stream throughout)
    for name in list(self._touched):
      a = np.asarray(self._cache[name], dtype=np.float64)
      if not p.view_semantics:
        dev = self._shadow.get(name)
        if dev is not None and name not in self._written and dev.shape == a.shape and np.array_equal(a, dev, equal_nan=True):
          continue      # only read since the last launch: the device already holds these values
      if name == 'xfrc_applied':
        # A READ marks an input as touched too (the caller may have written into the array it was handed).  Uploading
        # xfrc_applied switches the kernel's external-force path on for good (6 nbody reals per environment per step):
        # all-zero forces are only sent once a non-zero one has been, i.e. when there is something to clear.
        if not a.any() and not self._p.__dict__.get('_xfrc_sent', False):
          continue
        self._p._xfrc_sent = True
      send(name, a.reshape(p.batch_size, -1))
    self._touched.clear()
    self._written.clear()
    if not p.view_semantics:
      self._shadow.clear()
    if p.__dict__.pop('_needs_forward', False):      # a batch rebuilt by _push_model: see Physics._rebuild_batch
      warm = p.batch.get('qacc_warmstart')
      p.batch.forward(True)
      p.batch.set('qacc_warmstart', warm)

  def _invalidate(self):
    """After a launch every handed-out array is dropped: the next access fetches a fresh one.  (The reference's arrays
    are views into mjData that follow the simulation in place; here a held reference keeps the values it was read with,
    unless the Physics was built with `view_semantics`: then every handed-out array is the SAME ndarray for the life of
    the model, rewritten in place after each launch and watched for writes -- what dm_control.mjcf's bindings, which
    keep the arrays themselves, rely on.  Single environments only: the copies are per launch.)"""
    self._touched.clear()
    if self._reads:
      object.__setattr__(self, '_habit', tuple(sorted(self._reads)))
    self._reads.clear()
    self._prefetched.clear()
    if not self._p.view_semantics:
      self._cache.clear()
      self._shadow.clear()
      self._written.clear()
      return
    for name in list(self._cache):
      old = self._cache[name]
      try:
        new = self._fetch(name)
      except Exception:      # pylint: disable=broad-except
        del self._cache[name]
        continue
      if isinstance(old, np.ndarray) and old.ndim and old.shape == np.shape(new):
        np.copyto(old, new)
      else:
        self._cache[name] = new
      if name in _INPUT_FIELDS and isinstance(self._cache[name], np.ndarray) and self._cache[name].ndim:
        self._shadow[name] = np.array(self._cache[name], copy=True)

  @property
  def contact(self):
    """Active contacts of a single environment as a structured array with the
    reference's field names (wrapper/core.py:564-567)."""
    if self._p.batch_size != 1:
      raise NotImplementedError('data.contact is per environment; use contact_* fields for batches')
    n = int(self._get('ncon'))
    out = np.zeros(n, dtype=[('geom1', np.int32), ('geom2', np.int32), ('dist', np.float64),
                             ('pos', np.float64, 3), ('frame', np.float64, 9)])
    out['geom1'] = self._get('contact_geom1')[:n]
    out['geom2'] = self._get('contact_geom2')[:n]
    out['dist'] = self._get('contact_dist')[:n]
    out['pos'] = self._get('contact_pos').reshape(-1, 3)[:n]
    out['frame'] = self._get('contact_frame').reshape(-1, 9)[:n]
    # a record array: `contact.geom1` on the whole array AND on each element, as with mjData.contact
    # (locomotion/tasks/go_to_target.py:189-199 iterates it and reads `contact.geom1 / contact.geom2`)
    return out.view(np.recarray)

  def contact_force(self, contact_id):
    """(force, torque) of a contact in the contact frame, 2 x 3 (batch: B x 2 x 3), order
    (normal, tangent, tangent).  Like the reference (wrapper/core.py:527-552) this re-solves the
    constraints at the current state first (one mj_forward launch)."""
    p = self._p
    ncon = np.atleast_1d(self._get('ncon'))
    if not 0 <= contact_id < int(ncon.min()):
      raise ValueError('`contact_id` must be between 0 and {max_valid} (inclusive), got: {actual}.'
                       .format(max_valid=int(ncon.min()) - 1, actual=contact_id))
    self._upload()
    warm = p.batch.get('qacc_warmstart')
    p.batch.forward(False)
    p.batch.set('qacc_warmstart', warm)      # a query must not move the solver's warm start
    self._cache.clear()
    self._prefetched.clear()      # (read before this forward: stale now)
    w = p.batch.get('contact_force').reshape(p.batch_size, -1, 2, 3)[:, contact_id]
    return w[0] if p.batch_size == 1 else w

  def object_velocity(self, object_id, object_type, local_frame=False):
    """6D velocity (linear, angular) of a body / xbody / geom / site, 2 x 3 (batch: B x 2 x 3);
    mj_objectVelocity (wrapper/core.py:500-525) evaluated on the host from cvel / subtree_com."""
    p = self._p
    m = p.model
    kinds = {'body': ('body', 'xipos', None), 'xbody': ('body', 'xpos', 'xmat'), 'geom': ('geom', 'geom_xpos', 'geom_xmat'),
             'site': ('site', 'site_xpos', 'site_xmat'),
             mjcf_compiler.C['DMC_OBJ_BODY']: ('body', 'xipos', None), mjcf_compiler.C['DMC_OBJ_XBODY']: ('body', 'xpos', 'xmat'),
             mjcf_compiler.C['DMC_OBJ_GEOM']: ('geom', 'geom_xpos', 'geom_xmat'),
             mjcf_compiler.C['DMC_OBJ_SITE']: ('site', 'site_xpos', 'site_xmat')}
    if object_type not in kinds:
      raise ValueError('{!r} is not a valid object type for object_velocity'.format(object_type))
    kind, posf, matf = kinds[object_type]
    if not isinstance(object_id, (int, np.integer)):
      object_id = m.name2id(object_id, kind)
    B = p.batch_size
    body = {'body': object_id, 'geom': m.geom_bodyid[object_id] if kind == 'geom' else None,
            'site': m.site_bodyid[object_id] if kind == 'site' else None}[kind]
    pos = p.batch.get(posf).reshape(B, -1, 3)[:, object_id]
    if matf is None:     # inertial frame of a body: orientation xquat * body_iquat
      q = p.batch.get('xquat').reshape(B, -1, 4)[:, object_id]
      mat = np.stack([mjcf_compiler.quat_to_mat(mjcf_compiler.quat_mul(q[e], m.body_iquat[object_id])) for e in range(B)])
    else:
      mat = p.batch.get(matf).reshape(B, -1, 3, 3)[:, object_id]
    cvel = p.batch.get('cvel').reshape(B, -1, 6)[:, body]
    root = m.body_rootid[body]
    com = p.batch.get('subtree_com').reshape(B, -1, 3)[:, root]
    ang = cvel[:, :3]
    lin = cvel[:, 3:] - np.cross(pos - com, ang)
    if local_frame:
      ang = np.einsum('bij,bi->bj', mat, ang)
      lin = np.einsum('bij,bi->bj', mat, lin)
    out = np.stack([lin, ang], axis=1)
    return out[0] if B == 1 else out


class _Axis:
  """Row (or column) names -> indices; ragged rows span several entries."""

  def __init__(self, names, starts=None, sizes=None):
    self.names = list(names)
    self.starts = starts
    self.sizes = sizes
    self.lookup = {n: i for i, n in enumerate(self.names) if n}

  def convert(self, key):
    if isinstance(key, (str, bytes)):
      key = key.decode() if isinstance(key, bytes) else key
      if key not in self.lookup:
        raise KeyError(key)
      i = self.lookup[key]
      if self.starts is None:
        return i
      # ragged axes (qpos by joint, sensordata by sensor, ...) always yield a slice,
      # like the reference's RaggedNamedAxis: qpos['slider'] has shape (1,)
      return slice(int(self.starts[i]), int(self.starts[i] + self.sizes[i]))
    if isinstance(key, (list, np.ndarray)) and len(key):
      arr = np.asarray(key)
      if arr.size and isinstance(arr.flat[0], (str, bytes)):
        if self.starts is None:
          # an array of names of any shape -> the same shape of indices (mujoco/index.py:364-378): two such keys then
          # combine by numpy's rules -- paired element by element, or broadcast as in xpos[names.reshape(-1, 1), ['x', 'z']]
          return np.array([self.convert(k) for k in arr.flat], dtype=np.intp).reshape(arr.shape)
        idx = []      # ragged rows: the entries of every named element, flattened (index.py:430-442)
        for k in arr.flat:
          c = self.convert(k)
          idx.extend(range(c.start, c.stop))
        return idx
    return key


class FieldIndexer:
  """`physics.named.data.xpos['torso', 'z']`-style access (reference:
  dm_control/mujoco/index.py:455-560); batched arrays index their trailing axes."""

  def __init__(self, getter, row_axis, col_axis=None, batched=False, setter=None):
    self._get, self._rows, self._cols, self._batched, self._set = getter, row_axis, col_axis, batched, setter

  def _convert(self, key):
    if not isinstance(key, tuple):
      key = (key,)
    if len(key) > 2:
      raise IndexError('too many indices')
    out = [self._rows.convert(key[0])]
    if len(key) == 2:
      out.append(self._cols.convert(key[1]) if self._cols is not None else key[1])
    if self._batched:
      out = [slice(None)] + out
    return tuple(out)      # numpy's own indexing rules from here on, as in the reference (index.py:487-500)

  def __getitem__(self, key):
    return self._get()[self._convert(key)]

  def __setitem__(self, key, value):
    self._get()[self._convert(key)] = value
    if self._set:
      self._set()

  @property
  def axes(self):
    return self._rows, self._cols

  # the two private members dm_control.mjcf.physics.Binding reaches for (mjcf/physics.py:286-296)
  @property
  def _field(self):
    return self._get()

  def _convert_key(self, key):
    out = self._convert(key)
    return out if isinstance(key, tuple) else out[0]

  def __repr__(self):
    return 'FieldIndexer(rows=%r)' % (self._rows.names,)


class _Named:
  pass


def _make_axes(model):
  m = model
  jq = _Axis(m.names['joint'], m.jnt_qposadr, [{0: 7, 1: 4, 2: 1, 3: 1}[t] for t in m.jnt_type])
  jv = _Axis(m.names['joint'], m.jnt_dofadr, [{0: 6, 1: 3, 2: 1, 3: 1}[t] for t in m.jnt_type])
  return {
      'joint_q': jq, 'joint_v': jv,
      'actuator': _Axis(m.names['actuator']),
      # 'na': every actuator, sized by its number of activation states (0 or 1), as index.py:93-99 / :190-215 do
      'act': _Axis(m.names['actuator'],
                   np.concatenate([[0], np.cumsum(np.asarray(getattr(m, 'actuator_dyntype', np.zeros(0))) != 0)])[:-1],
                   (np.asarray(getattr(m, 'actuator_dyntype', np.zeros(0))) != 0).astype(int)),
      'sensor': _Axis(m.names['sensor'], m.sensor_adr, m.sensor_dim),
      'body': _Axis(m.names['body']), 'geom': _Axis(m.names['geom']), 'site': _Axis(m.names['site']),
      'joint': _Axis(m.names['joint']),
      'sensor_row': _Axis(m.names['sensor']), 'tendon': _Axis(m.names.get('tendon', [])),
      'light': _Axis(m.names.get('light', [])), 'material': _Axis(m.names.get('material', [])),
      'key': _Axis(m.names.get('key', [])),
      # mocap_pos / mocap_quat rows: the bodies with body_mocapid >= 0, in mocap-id order
      'mocap': _Axis([n for _, n in sorted((int(k), m.names['body'][b]) for b, k in enumerate(getattr(m, 'body_mocapid', ())) if k >= 0)]),
  }


# named.model exposes every array of the compiled model whose rows belong to named objects (mujoco/index.py:177-267
# derives the same from the sizes table): the row axis follows the field's prefix, columns are addressable by name for
# the fields of index.py:103-174 (_COLUMN_ID_TO_FIELDS).
_MODEL_PREFIX_AXES = (('body_', 'body'), ('jnt_', 'joint'), ('dof_', 'joint_v'), ('geom_', 'geom'), ('site_', 'site'),
                      ('actuator_', 'actuator'), ('sensor_', 'sensor_row'), ('tendon_', 'tendon'), ('light_', 'light'),
                      ('mat_', 'material'), ('key_', 'key'))
_MODEL_EXTRA_AXES = {'qpos0': 'joint_q', 'qpos_spring': 'joint_q'}
_XYZ_FIELDS = {'body_pos', 'body_ipos', 'body_inertia', 'jnt_pos', 'jnt_axis', 'geom_size', 'geom_pos', 'site_size',
               'site_pos', 'light_pos', 'light_dir'}
_QUAT_FIELDS = {'body_quat', 'body_iquat', 'geom_quat', 'site_quat'}
_RGBA_FIELDS = {'geom_rgba', 'site_rgba', 'mat_rgba'}


def _model_field_axes(model, axes):
  out = {}
  for name, value in vars(model).items():
    if not isinstance(value, np.ndarray) or value.ndim not in (1, 2):
      continue
    kind = _MODEL_EXTRA_AXES.get(name)
    if kind is None:
      kind = next((k for pre, k in _MODEL_PREFIX_AXES if name.startswith(pre)), None)
    if kind is None or kind not in axes:
      continue
    ax = axes[kind]
    nrows = (ax.starts[-1] + ax.sizes[-1] if len(ax.names) else 0) if ax.starts is not None else len(ax.names)
    if value.shape[0] != nrows:
      continue
    cols = None
    if value.ndim == 2:
      cols = (_Axis(_COLS[3]) if name in _XYZ_FIELDS and value.shape[1] == 3 else
              _Axis(_COLS[4]) if name in _QUAT_FIELDS and value.shape[1] == 4 else
              _Axis(['r', 'g', 'b', 'a']) if name in _RGBA_FIELDS and value.shape[1] == 4 else None)
    out[name] = (ax, cols)
  return out


class Physics(control.Physics):
  """Batched MuJoCo-semantics physics on one MI355X (see module docstring)."""

  _contexts = None
  view_semantics = False      # see _Data._invalidate; subclasses that serve dm_control.mjcf bindings switch it on

  def __init__(self, model, batch_size=1, device_id=0, precision=64, **batch_kwargs):
    """`precision`: 64 (default: drop-in numerics, tracks the CPU reference to
    rounding) or 32 (the throughput configuration BASELINE.json benchmarks)."""
    if not isinstance(model, mjcf_compiler.Model):
      raise TypeError('model must be a compiled Model; use Physics.from_xml_string')
    self.model = model
    self.batch_size = int(batch_size)
    self._batch_kwargs = dict(batch_kwargs, device_id=device_id)   # contact caps etc.: copies / pickles keep them
    self.batch = self._create_batch(model, device_id, precision, batch_kwargs)
    self.data = _Data(self)
    self._warnings_cause_exception = True
    self._warnings_seen = np.zeros((self.batch_size, len(_WARNING_NAMES)), dtype=np.int64)
    self._build_named()
    self._model_flat = None
    self._model_pushed = {f: np.array(getattr(model, f), dtype=np.float64, copy=True)
                          for f in _MUTABLE_MODEL_FIELDS if hasattr(model, f)}
    # Every other model array feeds tables that are derived once at batch creation (contact-pair mixing, inertias,
    # invweight0, ...): a write would be silently ignored by the device, so it is made to fail instead
    # (ValueError: assignment destination is read-only).  The writable ones are _MUTABLE_MODEL_FIELDS.
    # The freeze is applied to a PRIVATE shallow copy of the model whose immutable arrays are read-only views: the
    # caller's compiled Model stays writable (it may be edited and used to build another Physics, as the reference allows).
    import copy as _copy
    if not (getattr(model, '_frozen_private', False) and not np.asarray(model.body_mass).flags.writeable):      # (copy(share_model=True) hands this Physics' own frozen copy on)
      self.model = _copy.copy(model)
      self.model._frozen_private = True
      self.model.opt = _copy.copy(model.opt)      # option edits of this Physics (model.opt.*, model.disable) stay its own
      self.model.opt.gravity = np.array(model.opt.gravity, dtype=np.float64)
      for name, value in vars(model).items():
        if not isinstance(value, np.ndarray):
          continue
        if name in _MUTABLE_MODEL_FIELDS + _HOST_ONLY_MODEL_FIELDS + _STRUCTURAL_MODEL_FIELDS:
          setattr(self.model, name, np.array(value))      # this Physics' own copy: its edits must not reach another Physics built from the same Model
        else:
          view = value.view()
          view.setflags(write=False)
          setattr(self.model, name, view)
    self._opt_pushed = self._opt_snapshot()      # (the batch was created from these options)
    self._filter_pushed = self._filter_snapshot()
    self._reload_from_data(self.data)
    try:
      self.after_reset()
    except control.PhysicsError as e:
      # The model's own qpos0 may be a state nobody simulates -- PyMJCF compositions put every attached entity at the
      # origin until initialize_episode places them (soccer 2v2: 91 overlapping contacts).  MuJoCo's arena holds them
      # all; this backend's contact capacity is finite, so a capacity warning raised by the CONSTRUCTION-time
      # mj_forward is logged (the counters keep it) instead of making the model unloadable.  Later launches raise.
      if not set(str(e).split('raised: ')[-1].split(', ')) <= {'mjWARN_CONTACTFULL', 'mjWARN_CNSTRFULL'}:
        raise
      logging.getLogger(__name__).warning('at the model\'s qpos0: %s', e)

  # contact capacities tried, in order, when the caller names none: MuJoCo sizes its contact buffer from an arena (any
  # number of contacts a model can produce fits), a drop-in user never sets `nconmax` -- so the facade asks for a generous
  # cap first and settles for less only where the model's scratch would not fit in LDS.  (The throughput path,
  # BatchedPhysics / suite.load, keeps its tuned per-model caps: suite/common.py DEFAULT_CAPS.)
  _AUTO_NCONMAX = (64, 48, 32, 0)

  def _create_batch(self, model, device_id, precision, batch_kwargs):
    if 'nconmax' in batch_kwargs or self.batch_size > 64:
      return BatchedPhysics(model, self.batch_size, device_id=device_id, precision=precision, **batch_kwargs)
    for cap in self._AUTO_NCONMAX:
      try:
        return BatchedPhysics(model, self.batch_size, device_id=device_id, precision=precision, nconmax=cap, **batch_kwargs)
      except Exception as e:      # pylint: disable=broad-except
        if cap == 0 or 'does not fit' not in str(e):
          raise

  def _reload_from_data(self, data):
    """The hook engine.Physics calls whenever it (re)binds an mjData (engine.py:392-430): subclasses override it to
    reset what they cache per model (suite/quadruped.py:146-151 clears its sensor / hinge name caches here)."""
    del data

  _OPT_INTS = ('disableflags', 'iterations', 'ls_iterations')
  _OPT_REALS = ('timestep', 'tolerance', 'ls_tolerance')

  def _opt_snapshot(self):
    o = self.model.opt
    return tuple(int(getattr(o, n)) for n in self._OPT_INTS) + tuple(float(getattr(o, n)) for n in self._OPT_REALS) + \
        tuple(float(g) for g in o.gravity)

  def _filter_snapshot(self):
    return np.concatenate([np.asarray(getattr(self.model, f), dtype=np.int64).ravel() for f in _STRUCTURAL_MODEL_FIELDS])

  def _rebuild_batch(self):
    """A new device batch for the model as it now stands (its candidate pair list recomputed), holding the input state of
    the old one.  mj_step / mj_forward would simply see the new filter bits at their next collision pass; here the pair
    table is part of the batch's layout.  The derived arrays of the new batch are brought up by a forward pass at the end
    of the pending upload (Data._upload), with the solver warm start restored as copy() does -- so a legacy-order step
    (step2 first) that follows finds contact forces of the NEW filter where MuJoCo's would still be the old one's.  The
    warning counters restart with the batch."""
    model = self.model
    mjcf_compiler.candidate_pairs(model)
    for name in ('pair_geom1', 'pair_geom2'):
      getattr(model, name).setflags(write=False)
    kw = {k: v for k, v in self._batch_kwargs.items() if k != 'device_id'}
    old = self.batch
    kw.setdefault('nconmax', old.info()['nconmax'])      # (the contact arrays handed out keep their shape)
    try:
      batch = self._create_batch(model, self._batch_kwargs.get('device_id', 0), old.precision, kw)
    except Exception:      # pylint: disable=broad-except  (more pairs than before: that cap may no longer fit)
      del kw['nconmax']
      batch = self._create_batch(model, self._batch_kwargs.get('device_id', 0), old.precision, kw)
    for name in _INPUT_FIELDS:
      a = np.asarray(old.get(name), dtype=np.float64).reshape(self.batch_size, -1)
      if name == 'xfrc_applied' and not a.any():
        continue
      batch.set(name, a)
    self.batch = batch      # (created from the model's current arrays and options: nothing else to re-send)
    if self.__dict__.get('_profiling', False):
      batch.enable_profiling(True)      # (the step timers restart with the batch)
    close = getattr(old, 'close', None)
    if close:
      close()
    self._opt_pushed = self._opt_snapshot()
    for f in self._model_pushed:
      self._model_pushed[f] = np.array(getattr(model, f), dtype=np.float64, copy=True)
    self._model_flat = None
    self._needs_forward = True
    self._warnings_seen[:] = 0

  def _push_model(self):
    filt = self._filter_snapshot()
    if not np.array_equal(filt, self._filter_pushed):
      self._filter_pushed = filt
      self._rebuild_batch()
    # mjOption members tasks change at run time (engine.py:326-333 / entities/props/duplo/utils.py:68 model.disable(...),
    # opt.timestep, opt.gravity): sent to the device when they differ from what it holds
    snap = self._opt_snapshot()
    if snap != getattr(self, '_opt_pushed', None):
      old = getattr(self, '_opt_pushed', None)
      names = self._OPT_INTS + self._OPT_REALS + ('gravity_x', 'gravity_y', 'gravity_z')
      for k, (n, v) in enumerate(zip(names, snap)):
        if old is None or old[k] != v:
          self.batch.set_opt(n, v)
      self._opt_pushed = snap
    flat = np.concatenate([np.asarray(getattr(self.model, f), dtype=np.float64).ravel() for f in self._model_pushed])
    if self._model_flat is not None and flat.shape == self._model_flat.shape and np.array_equal(flat, self._model_flat, equal_nan=True):
      return      # (one comparison for all the writable model arrays: this runs before every launch)
    self._model_flat = flat
    for f, old in self._model_pushed.items():
      cur = np.asarray(getattr(self.model, f), dtype=np.float64)
      if cur.shape != old.shape or not np.array_equal(cur, old):
        self.batch.set_model_real(f, cur)
        self._model_pushed[f] = cur.copy()

  # -- construction -----------------------------------------------------------------
  @classmethod
  def from_model(cls, model, **kw):
    return cls(model, **kw)

  @classmethod
  def from_xml_string(cls, xml_string, assets=None, **kw):
    return cls(mjcf_compiler.compile_xml(xml_string, assets), **kw)

  @classmethod
  def from_xml_path(cls, file_path, **kw):
    with open(file_path) as f:
      return cls.from_xml_string(f.read(), **kw)

  def _build_named(self):
    axes = _make_axes(self.model)
    batched = self.batch_size > 1
    named = _Named()
    named.data = _Named()
    named.model = _Named()
    for field, (rowkind, ncol) in _FIELD_AXES.items():
      cols = _Axis(_COLS[ncol]) if ncol else None
      touch = (lambda f=field: self.data._touched.add(f)) if field in _INPUT_FIELDS else None
      setattr(named.data, field, FieldIndexer(lambda f=field: self.data._get(f), axes[rowkind], cols, batched, touch))
    for field, (ax, cols) in _model_field_axes(self.model, axes).items():
      setattr(named.model, field, FieldIndexer(lambda f=field: getattr(self.model, f), ax, cols, False))
    self.named = named

  # -- the step surface -----------------------------------------------------------------
  def set_control(self, control_values):
    self.data.ctrl = control_values

  def step(self, nstep=1):
    """Advances every environment by `nstep` substeps in one kernel launch."""
    with self.check_invalid_state():
      self.data._upload()
      self.batch.legacy_step = bool(self.legacy_step)
      self.batch.step(nstep)
      self.data._invalidate()

  def forward(self):
    with self.check_invalid_state():
      self.data._upload()
      self.batch.forward(False)
      self.data._invalidate()

  def reset(self, keyframe_id=None):
    """mj_resetData (+keyframe) then forward with actuation disabled."""
    if keyframe_id is not None and not 0 <= keyframe_id < self.model.nkey:
      raise ValueError('keyframe_id {} is out of range [0, {})'.format(keyframe_id, self.model.nkey))
    self.data._invalidate()
    self.batch.reset(keyframe_id=keyframe_id)
    self._warnings_seen[:] = 0
    self.after_reset()

  def after_reset(self):
    with self.check_invalid_state():
      self.data._upload()
      self.batch.forward(True)
      self.data._invalidate()

  @contextlib.contextmanager
  def check_invalid_state(self):
    """Raises PhysicsError (or logs, under suppress_physics_errors) if the
    enclosed launch incremented any warning counter (engine.py:345-368)."""
    yield
    # (the first device read after a launch: the fields the caller's loop reads come back in the same round trip)
    w = np.asarray(self.data._fetch_device('warning')).astype(np.int64)
    new = w > self._warnings_seen
    self._warnings_seen = w
    if new.any():
      names = [_WARNING_NAMES[i] for i in np.nonzero(new.any(axis=0))[0]]
      msg = _INVALID_PHYSICS_STATE.format(warning_names=', '.join(names))
      if self._warnings_cause_exception:
        raise control.PhysicsError(msg)
      logging.warning(msg)

  @contextlib.contextmanager
  def suppress_physics_errors(self):
    prev = self._warnings_cause_exception
    self._warnings_cause_exception = False
    try:
      yield
    finally:
      self._warnings_cause_exception = prev

  def check_divergence(self):
    pass

  def enable_profiling(self):
    """engine.py:135-137: switches the step timers on (`data.timer[0].duration / .number`); here every launch is
    bracketed by hipEvents on its stream (dmc_batch_enable_profiling)."""
    self._profiling = True
    self.batch.enable_profiling(True)

  # -- state -------------------------------------------------------------------------------
  # mjtState bits (mujoco 3.x), in the order mj_getState concatenates them
  _STATE_BITS = (('time', 1 << 0), ('qpos', 1 << 1), ('qvel', 1 << 2), ('act', 1 << 3), ('qacc_warmstart', 1 << 4),
                 ('ctrl', 1 << 5), ('qfrc_applied', 1 << 6), ('xfrc_applied', 1 << 7), ('eq_active', 1 << 8),
                 ('mocap_pos', 1 << 9), ('mocap_quat', 1 << 10), ('userdata', 1 << 11), ('plugin_state', 1 << 12))

  def _state_components(self, sig):
    if not isinstance(sig, (int, np.integer)) or sig <= 0 or sig >= (1 << 13):
      raise ValueError('invalid state signature: {!r}'.format(sig))
    out = []
    for name, bit in self._STATE_BITS:
      if not sig & bit:
        continue
      if name == 'eq_active':
        n = len(getattr(self.model, 'eq_active0', ()))
        out.append((name, n))
      elif name in ('mocap_pos', 'mocap_quat'):
        out.append((name, (3 if name == 'mocap_pos' else 4) * int(getattr(self.model, 'nmocap', 0))))
      elif name in ('userdata', 'plugin_state'):
        out.append((name, 0))            # no user data / plugins in a compiled Model
      else:
        out.append((name, int(np.asarray(self.data._get(name)).reshape(self.batch_size, -1).shape[1])))
    return out

  def get_state(self, sig=None):
    """sig None: concatenated [qpos, qvel, act] (engine.py:235-249); otherwise what `mj_getState(sig)` returns
    (engine.py:246-249): the components selected by the mjtState bits of `sig`, in bit order."""
    if sig is None:
      return np.concatenate([np.asarray(self.data.qpos), np.asarray(self.data.qvel), np.asarray(self.data.act)], axis=-1)
    parts = []
    for name, n in self._state_components(sig):
      if name == 'eq_active':
        a = np.tile(np.asarray(self.model.eq_active0, dtype=np.float64), (self.batch_size, 1))
      elif n == 0:
        a = np.zeros((self.batch_size, 0))
      else:
        a = np.asarray(self.data._get(name), dtype=np.float64).reshape(self.batch_size, -1)
      parts.append(a)
    out = np.concatenate(parts, axis=1) if parts else np.zeros((self.batch_size, 0))
    return out[0] if self.batch_size == 1 else out

  def set_state(self, physics_state, sig=None):
    s = np.asarray(physics_state, dtype=np.float64)
    if sig is None:
      nq, nv, na = self.model.nq, self.model.nv, self.model.na
      if s.shape[-1] != nq + nv + na:
        raise ValueError('Input physics state has shape {}. Expected {}.'.format(s.shape, (nq + nv + na,)))
      self.data.qpos = s[..., :nq]
      self.data.qvel = s[..., nq:nq + nv]
      if na:
        self.data.act = s[..., nq + nv:]
      return
    comps = self._state_components(sig)
    total = sum(n for _, n in comps)
    if s.shape[-1] != total:
      raise ValueError('Input physics state has shape {}. Expected {}.'.format(s.shape, (total,)))
    start = 0
    for name, n in comps:
      chunk = s[..., start:start + n]
      start += n
      if name == 'eq_active':
        if n and not np.array_equal(np.broadcast_to(chunk, (self.batch_size, n)) != 0,
                                    np.tile(np.asarray(self.model.eq_active0) != 0, (self.batch_size, 1))):
          raise ValueError('eq_active cannot be changed at run time on this backend')
      elif n:
        setattr(self.data, name, chunk[..., 0] if name == 'time' else chunk.reshape(np.shape(self.data._get(name))))

  def copy(self, share_model=False):
    """engine.py:287-304: an independent Physics in the same state; the model is copied unless share_model."""
    import copy as _copy
    model = self.model if share_model else _copy.deepcopy(self.model)
    if not share_model:
      model.__dict__.pop('_frozen_private', None)      # an independent model: frozen afresh by the new Physics
    other = type(self)(model, batch_size=self.batch_size, precision=self.batch.precision, **self._batch_kwargs)
    # per-episode state that suite Physics subclasses keep on the instance (reacher / finger / manipulator targets)
    for k, v in vars(self).items():
      if not k.startswith('_') and k not in ('model', 'batch', 'data', 'named', 'batch_size'):
        setattr(other, k, _copy.deepcopy(v))
    for name in _INPUT_FIELDS:
      a = np.asarray(self.data._get(name), dtype=np.float64).reshape(self.batch_size, -1)
      if name == 'xfrc_applied':
        other._xfrc_sent = bool(a.any())
        if not a.any():
          continue      # (a fresh batch holds zeros; sending them would switch its external-force path on for good)
      other.batch.set(name, a)
    other.legacy_step = self.legacy_step
    other.data._invalidate()
    # refresh derived arrays, then restore the solver warm start that forward()
    # overwrote, so that the copy continues bit-identically (engine_test.py:549-572)
    other.batch.forward(True)
    other.batch.set('qacc_warmstart', np.asarray(self.data._get('qacc_warmstart'), dtype=np.float64).reshape(self.batch_size, -1))
    other._warnings_seen = other.batch.get('warning').astype(np.int64)
    return other

  __copy__ = copy

  def __deepcopy__(self, memo):
    del memo
    return self.copy()

  # pickling (engine_test.py:549-572): the compiled model plus the input state; the device batch
  # is rebuilt on load and the derived arrays recomputed, exactly as copy() does
  def __getstate__(self):
    self.data._upload()
    return dict(cls_model=self.model, batch_size=self.batch_size, precision=self.batch.precision,
                legacy_step=self.legacy_step, batch_kwargs=self._batch_kwargs,
                fields={n: self.batch.get(n) for n in _INPUT_FIELDS},
                attrs={k: v for k, v in vars(self).items()
                       if not k.startswith('_') and k not in ('model', 'batch', 'data', 'named', 'batch_size', 'legacy_step')})

  def __setstate__(self, st):
    Physics.__init__(self, st['cls_model'], batch_size=st['batch_size'], precision=st['precision'],
                     **st.get('batch_kwargs', {}))
    for n, v in st['fields'].items():
      if n == 'xfrc_applied':
        self._xfrc_sent = bool(np.asarray(v).any())
        if not self._xfrc_sent:
          continue
      self.batch.set(n, v)
    self.legacy_step = st['legacy_step']
    for k, v in st.get('attrs', {}).items():
      setattr(self, k, v)
    self.data._invalidate()
    self.batch.forward(True)
    self.batch.set('qacc_warmstart', st['fields']['qacc_warmstart'])
    self._warnings_seen = self.batch.get('warning').astype(np.int64)

  def reload_from_xml_string(self, xml_string, assets=None):
    """Swaps in a new model, keeping this object (engine.py:500-515): composer environments do this at
    every episode when their MJCF changed (composer/environment.py:377-383).  The device batch is rebuilt
    with the same batch size, precision and caps; the state is that of a freshly constructed Physics.
    Recompiling an unchanged XML costs a hash lookup (`mjcf_compiler.compile_xml` cache)."""
    model = mjcf_compiler.compile_xml(xml_string, assets)
    kwargs = dict(self._batch_kwargs)
    batch_size, precision, legacy = self.batch_size, self.batch.precision, self.legacy_step
    self.free()
    Physics.__init__(self, model, batch_size=batch_size, precision=precision, **kwargs)
    self.legacy_step = legacy

  def reload_from_xml_path(self, file_path):
    with open(file_path) as f:
      self.reload_from_xml_string(f.read())

  def render(self, *args, **kwargs):
    """Rendering (engine.py:178-233) is outside this backend's scope: there is no GL context here."""
    raise NotImplementedError('rendering is not part of the MI355X physics backend (DESIGN.md, out of scope)')

  def free(self):
    if getattr(self, 'batch', None) is not None:
      self.batch.close()
      self.batch = None

  def __del__(self):
    try:
      self.free()
    except Exception:  # pylint: disable=broad-except
      pass

  # -- accessors (engine.py:589-622) ---------------------------------------------------------
  def timestep(self):
    return self.model.opt.timestep

  def time(self):
    return self.data.time

  def control(self):
    return np.array(self.data.ctrl)

  def activation(self):
    return np.array(self.data.act)

  def state(self):
    return self.get_state()

  def position(self):
    return np.array(self.data.qpos)

  def velocity(self):
    return np.array(self.data.qvel)


def action_spec(physics):
  """BoundedArray over the actuator control ranges; unlimited actuators get
  +-mjMAXVAL (engine.py:1093-1103)."""
  m = physics.model
  limited = m.actuator_ctrllimited.ravel().astype(bool)
  lo = np.where(limited, m.actuator_ctrlrange[:, 0], -mjMAXVAL)
  hi = np.where(limited, m.actuator_ctrlrange[:, 1], mjMAXVAL)
  shape = (m.nu,) if physics.batch_size == 1 else (physics.batch_size, m.nu)
  return specs.BoundedArray(shape=shape, dtype=float, minimum=np.broadcast_to(lo, shape),
                            maximum=np.broadcast_to(hi, shape))
