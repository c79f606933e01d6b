"""Device-resident physics for composer-style tasks: the batch's SoA fields as torch tensors, and `bind()`.

The reference's composer tasks never index mjData by hand: they go through `mjcf.Physics.bind(elements)`
(dm_control/mjcf/physics.py:516-652), which resolves MJCF elements to rows of the mjData / mjModel arrays once and
then exposes them as attributes (`physics.bind(actuators).ctrl = action`, `physics.bind(root_body).xpos`).
`DevicePhysics.bind(kind, names)` is that resolution against a compiled model, over the (rows, B) SoA tensors of a
HIP batch: attribute reads are gathers of shape (n_elements, width, B) (width dropped when 1), attribute writes are
scatters that mark the derived arrays stale, exactly the attribute table of the reference
(`mjcf/physics.py:_ATTRIBUTES`, the subset the locomotion walkers and tasks use).

torch is plumbing here (device memory + elementwise task arithmetic); every physics quantity comes out of the
fused HIP kernel through zero-copy bound tensors (`dmc_batch_bind`).
"""
import numpy as np

from dm_control_amd import mjcf_compiler

from dm_control_amd.batch import BatchedPhysics, OUT

# mjData fields a task may read, the output bit that makes the kernel write them, and whether they are state
_DERIVED = {
    'sensordata': 'sensor', 'xpos': 'xpos', 'xquat': 'xquat', 'xmat': 'xmat', 'xipos': 'xipos',
    'subtree_com': 'subtree_com', 'geom_xpos': 'geom', 'geom_xmat': 'geom', 'site_xpos': 'site', 'site_xmat': 'site',
    'qacc': 'qacc', 'actuator_force': 'actuator', 'qfrc_actuator': 'qfrc', 'cvel': 'cvel',
    'contact_geom1': 'contact_ids', 'contact_geom2': 'contact_ids',
}
_INT_FIELDS = ('ncon', 'nefc', 'solver_iter', 'contact_geom1', 'contact_geom2')

# kind -> attribute -> (field, width) with rows = width * element_id + k; joints and sensors are ragged
_ATTRIBUTES = {
    'body': {'xpos': ('xpos', 3), 'xquat': ('xquat', 4), 'xmat': ('xmat', 9), 'xipos': ('xipos', 3),
             'subtree_com': ('subtree_com', 3), 'cvel': ('cvel', 6)},
    'geom': {'xpos': ('geom_xpos', 3), 'xmat': ('geom_xmat', 9)},
    'site': {'xpos': ('site_xpos', 3), 'xmat': ('site_xmat', 9)},
    'actuator': {'ctrl': ('ctrl', 1), 'force': ('actuator_force', 1)},
    'joint': {'qpos': ('qpos', 'q'), 'qvel': ('qvel', 'v'), 'qacc': ('qacc', 'v'), 'qfrc_actuator': ('qfrc_actuator', 'v')},
    'sensor': {'sensordata': ('sensordata', 's')},
}
# model constants reachable through a binding (read-only here: shared by the batch; per-env values go through
# DevicePhysics.set_geom_pos / set_geom_size overrides, see dmc_batch per-env model deltas)
_MODEL_ATTRIBUTES = {
    'body': {'pos': 'body_pos', 'quat': 'body_quat', 'mass': 'body_mass'},
    'geom': {'pos': 'geom_pos', 'quat': 'geom_quat', 'size': 'geom_size', 'friction': 'geom_friction'},
    'site': {'pos': 'site_pos', 'quat': 'site_quat', 'size': 'site_size'},
    'joint': {'range': 'jnt_range'},
    'actuator': {'ctrlrange': 'actuator_ctrlrange', 'gear': 'actuator_gear'},
}
_JNT_QW = {0: 7, 1: 4, 2: 1, 3: 1}
_JNT_VW = {0: 6, 1: 3, 2: 1, 3: 1}


class Binding:
  """What `mjcf.Physics.bind(elements)` returns (mjcf/physics.py:243-420), for a batch."""

  def __init__(self, physics, kind, names):
    if kind not in _ATTRIBUTES:
      raise ValueError('elements of type %r cannot be bound to physics' % kind)
    object.__setattr__(self, '_physics', physics)
    object.__setattr__(self, '_kind', kind)
    single = isinstance(names, str)
    object.__setattr__(self, '_single', single)
    names = [names] if single else list(names)
    m = physics.model
    object.__setattr__(self, '_names', names)
    object.__setattr__(self, '_ids', [m.name2id(n, kind) for n in names])
    object.__setattr__(self, '_cache', {})

  @property
  def element_id(self):
    ids = self._ids
    return ids[0] if self._single else np.asarray(ids)

  def _rows(self, attr):
    """(index tensor of rows, per-element width or None when ragged) for a data attribute."""
    c = self._cache.get(attr)
    if c is not None:
      return c
    table = _ATTRIBUTES[self._kind]
    if attr not in table:
      raise AttributeError('bound element <%s> does not have attribute %r' % (self._kind, attr))
    field, width = table[attr]
    m = self._physics.model
    rows, w = [], width
    for i in self._ids:
      if width == 'q':
        a, n = int(m.jnt_qposadr[i]), _JNT_QW[int(m.jnt_type[i])]
      elif width == 'v':
        a, n = int(m.jnt_dofadr[i]), _JNT_VW[int(m.jnt_type[i])]
      elif width == 's':
        a, n = int(m.sensor_adr[i]), int(m.sensor_dim[i])
      else:
        a, n = width * i, width
      rows.extend(range(a, a + n))
    if not isinstance(width, int):      # ragged: per-element reshape only when every element has the same width
      per = [(_JNT_QW if width == 'q' else _JNT_VW)[int(m.jnt_type[i])] if width in 'qv' else int(m.sensor_dim[i]) for i in self._ids]
      sizes = set(per)
      w = per[0] if len(sizes) == 1 else None
    torch = self._physics.torch
    idx = torch.tensor(rows, dtype=torch.int64, device=self._physics.device)
    self._cache[attr] = (field, idx, w)
    return self._cache[attr]

  def __getattr__(self, attr):
    if attr.startswith('_'):
      raise AttributeError(attr)
    kind = self._kind
    if attr in _MODEL_ATTRIBUTES.get(kind, {}):
      arr = np.asarray(getattr(self._physics.model, _MODEL_ATTRIBUTES[kind][attr]))
      arr = arr.reshape(arr.shape[0], -1)[self._ids]
      return arr[0] if self._single else arr
    field, idx, w = self._rows(attr)
    t = self._physics.field(field).index_select(0, idx)          # (rows, B)
    n = len(self._ids)
    if w is not None and w > 1:
      t = t.reshape(n, w, -1)
      return t[0] if self._single else t
    if w == 1:
      return t[0] if self._single else t
    return t                                                     # ragged: flat (rows, B)

  def __setattr__(self, attr, value):
    field, idx, _ = self._rows(attr)
    if field not in ('ctrl', 'qpos', 'qvel'):
      raise AttributeError('attribute %r of bound <%s> elements is computed by the engine and cannot be written' % (attr, self._kind))
    torch = self._physics.torch
    dst = self._physics.field(field)
    v = torch.as_tensor(value, dtype=dst.dtype, device=dst.device)
    if v.dim() == 0:
      v = v.expand(idx.numel(), dst.shape[1])
    elif v.dim() == 1:
      v = v[:, None].expand(idx.numel(), dst.shape[1])
    dst.index_copy_(0, idx, v.reshape(idx.numel(), -1).expand(idx.numel(), dst.shape[1]))
    if field != 'ctrl':
      self._physics.mark_as_dirty()


class DevicePhysics:
  """One HIP batch whose fields are torch tensors (zero copy), with the `mjcf.Physics` conveniences tasks use."""

  def __init__(self, model, batch_size, device_id=0, precision=32, outputs=('sensordata', 'xpos', 'xmat'), **caps):
    import torch
    self.torch = torch
    self.model = model
    self.B = int(batch_size)
    self.device = torch.device('cuda', device_id)
    self.dtype = torch.float32 if precision == 32 else torch.float64
    self.batch = BatchedPhysics(model, self.B, device_id=device_id, precision=precision, **caps)
    self._fields = {}
    self._gathers = {}
    self._consts = {}
    mask = 0
    names = ['qpos', 'qvel', 'ctrl', 'qacc_warmstart', 'time', 'ncon', 'warning', 'env_mode'] + (['act'] if model.na else [])
    for f in outputs:
      if f not in _DERIVED:
        raise ValueError('unknown derived field %r' % f)
      mask |= OUT[_DERIVED[f]]
      names.append(f)
      if f == 'contact_geom1':
        names.append('contact_geom2')
    self.batch.set_output_mask(mask)
    for name in dict.fromkeys(names):
      rows, is_int = self.batch._rows(name)
      if name == 'time':
        dt = torch.float64
      elif is_int:
        dt = torch.int32
      else:
        dt = self.dtype
      t = torch.zeros((max(rows, 1), self.B), dtype=dt, device=self.device)
      self.batch.bind(name, t.data_ptr())
      self._fields[name] = t
    self._dirty = True
    # dmc_batch_step refuses legacy_step 2 (the step launch that ends with mj_forward) for RK4 models: an
    # observation_forward task on such a model takes the separate forward launch (composer/environment.py)
    self.supports_forward_after = int(model.opt.integrator) != mjcf_compiler.C['DMC_INT_RK4']
    self.reset()

  # -- fields ------------------------------------------------------------------------------------
  def field(self, name):
    try:
      return self._fields[name]
    except KeyError:
      raise KeyError('field %r is not bound; list it in `outputs` when creating the physics' % name) from None

  def __getattr__(self, name):
    f = self.__dict__.get('_fields', {})
    if name in f:
      return f[name]
    raise AttributeError(name)

  def bind(self, kind, names):
    """physics.bind(elements) of the reference, elements given as (kind, name or names)."""
    return Binding(self, kind, names)

  def declare_env_geoms(self, names):
    """Per-environment model deltas (dmc_batch_set_env_geoms): the world-fixed geoms `names` get their pose and
    size from the (16 n, B) tensor field 'env_geom' -- rows of geom k: pos 3, xmat 9 (row-major), size 3, bounding
    radius 1 -- initialised from the model.  What the reference does by editing the MJCF and recompiling."""
    torch = self.torch
    self.batch.set_env_geoms(names)
    init = self.batch.get('env_geom')                     # (B, 16 n) host
    t = torch.from_numpy(np.ascontiguousarray(init.T)).to(self.device).to(self.dtype).contiguous()
    self.batch.sync()
    self.batch.bind('env_geom', t.data_ptr())
    self._fields['env_geom'] = t
    self.env_geoms = list(names)
    return t

  def const(self, values):
    """Device tensor of a small host constant, uploaded ONCE per distinct value: tasks call this inside their hooks, and
    a host-to-device copy inside a captured HIP graph is not allowed (nor wanted on every control step)."""
    a = np.asarray(values, dtype=np.float64)
    key = (a.shape, a.tobytes())
    t = self._consts.get(key)
    if t is None:
      t = self._consts[key] = self.torch.from_numpy(a).to(self.device).to(self.dtype)
    return t

  def stream(self):
    return self.torch.cuda.current_stream().cuda_stream

  def gather(self, table):
    """(B, table.size) observation matrix of an `observation.GatherTable`: one launch of the gather kernel."""
    g = self._gathers.get(id(table))
    if g is None:
      g = self._gathers[id(table)] = (table, table.on_device(self.batch))
    return g[1](stream=self.stream())

  # -- control.Physics surface (rl/control.py:206-267) ----------------------------------------------
  def mark_as_dirty(self):
    """State was edited through a tensor: derived arrays are stale until the next forward / step."""
    self._dirty = True
    self.batch.invalidate(stream=self.stream())      # on the current stream: part of a captured control step

  def forward(self, disable_actuation=False):
    self.batch.forward(disable_actuation=disable_actuation, stream=self.stream())
    self._dirty = False

  def step(self, nstep=1, forward_after=False):
    self.batch.step(nstep, stream=self.stream(), forward_after=forward_after)
    self._dirty = False

  def substep_probe(self, geom_name, capacity):
    """(capacity, 3, B) tensor that every step launch fills with the world position of `geom_name` after each of its
    physics steps (dmc_batch_set_step_probe): position-only after_substep hooks read it after ONE fused launch."""
    key = ('probe', geom_name)
    t = self._consts.get(key)
    if t is None or t.shape[0] < capacity:
      t = self._consts[key] = self.torch.zeros((capacity, 3, self.B), dtype=self.dtype, device=self.device)
      self.batch.set_step_probe(self.model.name2id(geom_name, 'geom'), t.data_ptr(), capacity)
    return t

  def step1(self):
    self.batch.step1(stream=self.stream())

  def step2(self):
    self.batch.step2(stream=self.stream())

  def timestep(self):
    return float(self.model.opt.timestep)

  def reset(self, mask=None):
    """mj_resetData for the masked environments (all when None), on device: qpos0, zero velocity / control /
    activation / warm start / time; derived arrays refreshed by the caller's forward (reset_context order)."""
    torch = self.torch
    f = self._fields
    q0 = self.const(self.model.qpos0)[:, None]
    if mask is None:
      f['qpos'].copy_(q0.expand_as(f['qpos']))
      for n in ('qvel', 'ctrl', 'qacc_warmstart', 'time') + (('act',) if 'act' in f else ()):
        f[n].zero_()
    else:
      m2 = mask[None, :]
      f['qpos'].copy_(torch.where(m2, q0, f['qpos']))
      for n in ('qvel', 'ctrl', 'qacc_warmstart', 'time') + (('act',) if 'act' in f else ()):
        f[n].masked_fill_(m2, 0)      # (one in-place kernel per field: this runs in every control step, under the restart mask)
    self.mark_as_dirty()

  def close(self):
    self.batch.close()
