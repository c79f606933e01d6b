"""Named-field observations compiled to a gather table.

The reference builds observations from `MJCFFeature(kind, elements)` observables
(dm_control/composer/observation/observable/mjcf.py:43): "this field of mjData, for these named elements",
evaluated per environment in Python by the observation updater
(composer/observation/updater.py:120-331).  Here the same description is resolved ONCE against the compiled
model into flat row indices of the batch's field arrays -- every mjData field of a batch is a (rows, B) SoA
array on the device and a (B, rows) array on the host mirror -- so evaluating an observation is one indexed
gather per field and works for the whole batch at once.

  table = GatherTable(model, [('qpos', ['bthigh', 'bshin']), ('xpos', ['torso'], 'z'), ('sensordata', None)])
  obs = table.gather(physics)             # (B, table.size) float64 (B == 1: (table.size,))
  rows = table.rows['xpos']               # the row indices, e.g. to index a device tensor of that field

On the device the same table is ONE launch of the library's gather kernel (include/dmc_batch.h dmc_gather_*):

  dg = table.on_device(batch)             # a BatchedPhysics
  obs = dg(out=None, stream=None)         # (B, table.size) torch tensor in batch precision, env-major

An entry may carry a corruptor, the per-value transform some of the reference's observables apply
(`corruptor=` of MJCFFeature): ('sensordata', touch_sensor_names, None, ('greater', 1e-3)).
"""
import collections
import ctypes

import numpy as np

# corruptors the gather kernel implements (csrc/dmc_api.hip GOP_*): name -> (op code, host function of (v, p))
OPS = {
    None: (0, lambda v, p: v),
    'greater': (1, lambda v, p: (v > p).astype(np.float64)),      # walkers' touch sensors (legacy_base.py:262-265)
    'tanh2': (2, lambda v, p: np.tanh(2 * v / p)),                # torque sensors (cmu_humanoid.py:462-465)
    'log1p': (3, lambda v, p: np.log1p(v)),
    'asinh': (4, lambda v, p: np.arcsinh(v)),
}

# field -> (kind of named object, entries per object, column names)
_FIELDS = {
    'qpos': ('joint_q', None, None), 'qvel': ('joint_v', None, None), 'qacc': ('joint_v', None, None),
    'ctrl': ('actuator', 1, None), 'actuator_force': ('actuator', 1, None), 'act': (None, 1, None),
    'sensordata': ('sensor', None, None),
    'xpos': ('body', 3, 'xyz'), 'xipos': ('body', 3, 'xyz'), 'subtree_com': ('body', 3, 'xyz'),
    'xquat': ('body', 4, ('qw', 'qx', 'qy', 'qz')),
    'xmat': ('body', 9, ('xx', 'xy', 'xz', 'yx', 'yy', 'yz', 'zx', 'zy', 'zz')),
    'geom_xpos': ('geom', 3, 'xyz'), 'geom_xmat': ('geom', 9, ('xx', 'xy', 'xz', 'yx', 'yy', 'yz', 'zx', 'zy', 'zz')),
    'site_xpos': ('site', 3, 'xyz'), 'site_xmat': ('site', 9, ('xx', 'xy', 'xz', 'yx', 'yy', 'yz', 'zx', 'zy', 'zz')),
}


def _spans(model, kind):
  """name -> (first row, number of rows) for ragged kinds (joints, sensors)."""
  m = model
  if kind == 'joint_q':
    return {n: (int(m.jnt_qposadr[i]), {0: 7, 1: 4, 2: 1, 3: 1}[int(m.jnt_type[i])]) for i, n in enumerate(m.names['joint']) if n}
  if kind == 'joint_v':
    return {n: (int(m.jnt_dofadr[i]), {0: 6, 1: 3, 2: 1, 3: 1}[int(m.jnt_type[i])]) for i, n in enumerate(m.names['joint']) if n}
  if kind == 'sensor':
    return {n: (int(m.sensor_adr[i]), int(m.sensor_dim[i])) for i, n in enumerate(m.names['sensor']) if n}
  raise KeyError(kind)


class GatherTable:
  """Resolves [(field, names[, columns])] against a compiled model.  `names` None = every row of the field;
  `columns` (fixed-width fields only) = a string / list of column names ('z', ['zx', 'zy', 'zz'])."""

  def __init__(self, model, entries):
    self.model = model
    self.rows = collections.OrderedDict()
    self.slices = []                      # (field, first output column, count), in entry order
    self.flat = []                        # per output column: (field, row, op name, op parameter)
    out = 0
    for entry in entries:
      field, names = entry[0], entry[1]
      columns = entry[2] if len(entry) > 2 else None
      op, prm = entry[3] if len(entry) > 3 and entry[3] else (None, 0.0)
      if op not in OPS:
        raise ValueError('unknown corruptor %r' % (op,))
      if field not in _FIELDS:
        raise ValueError('field %r cannot be observed through a GatherTable' % field)
      kind, width, colnames = _FIELDS[field]
      idx = self._resolve(field, kind, width, colnames, names, columns)
      self.rows.setdefault(field, [])
      self.rows[field].extend(idx)
      self.slices.append((field, out, len(idx)))
      self.flat.extend((field, int(r), op, float(prm)) for r in idx)
      out += len(idx)
    self.size = out
    # per field: the rows to fetch and where each entry's values go in the flat observation
    self._plan = []
    cursor = {f: 0 for f in self.rows}
    for field, first, count in self.slices:
      self._plan.append((field, cursor[field], first, count))
      cursor[field] += count
    self.rows = collections.OrderedDict((f, np.asarray(r, dtype=np.int64)) for f, r in self.rows.items())

  def _resolve(self, field, kind, width, colnames, names, columns):
    m = self.model
    if width is None:                     # ragged: joints / sensors
      if columns is not None:
        raise ValueError('%s has no named columns' % field)
      spans = _spans(m, kind)
      if names is None:
        total = {'joint_q': m.nq, 'joint_v': m.nv, 'sensor': m.nsensordata}[kind]
        return list(range(total))
      idx = []
      for n in names:
        if n not in spans:
          raise KeyError('%s: no %s named %r' % (field, kind.split('_')[0], n))
        idx.extend(range(spans[n][0], spans[n][0] + spans[n][1]))
      return idx
    if kind is None:                      # act: unnamed rows
      return list(range(m.na)) if names is None else [int(i) for i in names]
    all_names = m.names[kind]
    objs = range(len(all_names)) if names is None else [m.name2id(n, kind) for n in names]
    if columns is None:
      cols = range(width)
    else:
      if colnames is None:
        raise ValueError('%s has no named columns' % field)
      wanted = [columns] if (isinstance(columns, str) and columns in colnames) else list(columns)
      cols = [list(colnames).index(c) for c in wanted]
    return [width * o + c for o in objs for c in cols]

  def gather(self, physics):
    """Evaluates the table on a `Physics` (host mirror): (B, size), or (size,) for a single environment."""
    B = physics.batch_size
    fetched = {f: np.asarray(physics.batch.get(f))[:, r] for f, r in self.rows.items()}
    out = np.zeros((B, self.size))
    for field, src, dst, count in self._plan:
      out[:, dst:dst + count] = fetched[field][:, src:src + count]
    for k, (_, _, op, prm) in enumerate(self.flat):
      if op is not None:
        out[:, k] = OPS[op][1](out[:, k], prm)
    return out[0] if B == 1 else out

  def on_device(self, batch):
    """The table as a device gather over `batch` (a BatchedPhysics)."""
    return DeviceGather(self, batch)


class DeviceGather:
  """One HIP launch per evaluation: (B, size) env-major observation matrix in batch precision."""

  def __init__(self, table, batch):
    from dm_control_amd import _native
    self._native = _native
    self.table, self.batch = table, batch
    n = table.size
    names = (ctypes.c_char_p * n)(*[f.encode() for f, _, _, _ in table.flat])
    rows = np.array([r for _, r, _, _ in table.flat], dtype=np.int32)
    ops = np.array([OPS[o][0] for _, _, o, _ in table.flat], dtype=np.int32)
    prm = np.array([p for _, _, _, p in table.flat], dtype=np.float64)
    self._ptr = ctypes.c_void_p()
    _native.check(_native.lib().dmc_gather_create(batch._ptr, n, names, rows.ctypes.data, ops.ctypes.data,
                                                  prm.ctypes.data, ctypes.byref(self._ptr)))

  def __call__(self, out=None, stream=None):
    import torch
    B = self.batch.batch_size
    if out is None:
      dt = torch.float32 if self.batch.precision == 32 else torch.float64
      out = torch.empty((B, self.table.size), dtype=dt, device=torch.device('cuda', self.batch.device_id))
    if stream is None:
      stream = torch.cuda.current_stream().cuda_stream
    self._native.check(self._native.lib().dmc_gather_run(self._ptr, out.data_ptr(), stream))
    return out

  def close(self):
    if self._ptr:
      self._native.lib().dmc_gather_destroy(self._ptr)
      self._ptr = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass
