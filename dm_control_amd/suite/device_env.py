"""Every suite task as a device-resident batched environment (completed).

`suite/torch_env.py` holds hand-written torch task layers for the six domains of the BASELINE configs and their
neighbours (14 tasks, per-environment auto-reset with randomisation drawn on the device).  This module serves the other
31 tasks -- and, being generic, all 45 -- WITHOUT a second copy of their task code: the host ports' own
`get_observation / get_reward` (dm_control_amd/suite/<domain>.py, the numpy restatements of
dm_control/suite/<domain>.py held bit-identical to the reference's modules by tests/test_reference_suite_domains.py) are
executed on DEVICE tensors.

How.  The HIP batch's fields are rebound to torch tensors (zero copy, `dmc_batch_bind`).  `TArr` wraps such a tensor
behind numpy's dispatch protocols (`__array_ufunc__`, `__array_function__`): `np.concatenate`, `np.where`,
`np.linalg.norm`, `np.exp`, fancy indexing, reductions ... called by the task code run as the corresponding torch
operation on the GPU and return `TArr`s; nothing is copied to the host (a `TArr` that does get converted -- `np.asarray`,
`float()` -- counts it in `TArr.host_reads`, which the tests hold at zero for every task).  A `DeviceView` -- an instance
of the domain's own `Physics` subclass, so that its accessors (`physics.horizontal()`, `physics.mouth_to_target()` ...)
are the host port's -- serves `data.*` and `named.data.*` from those tensors with the facade's named axes.

The control step (ctrl write, the fused physics launch, observation, reward) has no host decision in it, so it is captured
once into a HIP graph (torch.cuda.CUDAGraph; the library's kernel is launched on the capturing stream) and replayed:
the numpy-dispatch layer runs at capture time only.  Suite episodes end on the time limit alone, so all environments of
a batch restart together: the restart runs the host port's `initialize_episode` through the facade (host RNG, as
`suite.load` does; a few launches every 1000 steps) and refreshes the per-episode device copies of what it set on the
physics (targets ...) in place.

torch is plumbing here (memory + elementwise ops); the physics is the fused HIP kernel.
"""
import collections

import numpy as np


class TArr:
  """A torch tensor that numpy code can compute with (see the module docstring).  Batch-first shapes, like the facade's
  batched arrays."""

  __array_priority__ = 1000
  host_reads = 0            # conversions to host values (np.asarray / float / bool / iteration over 0-d): syncs
  default_float = None      # torch dtype python floats take when no tensor operand fixes it

  def __init__(self, t):
    self.t = t

  # -- array attributes -------------------------------------------------------------------------------------------
  @property
  def shape(self):
    return tuple(self.t.shape)

  @property
  def ndim(self):
    return self.t.dim()

  @property
  def size(self):
    return self.t.numel()

  @property
  def dtype(self):
    import torch
    return np.dtype({torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64,
                     torch.bool: np.bool_}[self.t.dtype])

  @property
  def T(self):
    return TArr(self.t.T)

  def __len__(self):
    return self.t.shape[0]

  def __iter__(self):
    return (TArr(self.t[i]) for i in range(self.t.shape[0]))

  def __repr__(self):
    return 'TArr(%r)' % (self.t,)

  # -- host conversions (counted) ----------------------------------------------------------------------------------
  def __array__(self, dtype=None, copy=None):
    TArr.host_reads += 1
    a = self.t.detach().cpu().numpy()
    return a.astype(dtype) if dtype is not None else a

  def __float__(self):
    TArr.host_reads += 1
    return float(self.t)

  def __bool__(self):
    TArr.host_reads += 1
    return bool(self.t)

  def __int__(self):
    TArr.host_reads += 1
    return int(self.t)

  # -- indexing ------------------------------------------------------------------------------------------------------
  def _key(self, key):
    import torch

    def one(k):
      if isinstance(k, TArr):
        return k.t
      if isinstance(k, np.ndarray):
        return _const(k, self.t.device)
      if isinstance(k, (list, tuple)) and len(k) and not isinstance(k[0], (slice, type(Ellipsis), type(None))):
        a = np.asarray(k)
        if a.dtype.kind in 'iub':
          return _const(a, self.t.device)
      if isinstance(k, np.integer):
        return int(k)
      return k
    if isinstance(key, tuple):
      return tuple(one(k) for k in key)
    return one(key)

  def __getitem__(self, key):
    return TArr(self.t[self._key(key)])

  def __setitem__(self, key, value):
    self.t[self._key(key)] = _tensor(value, self.t)

  # -- arithmetic ------------------------------------------------------------------------------------------------------
  def _bin(self, other, fn, reverse=False):
    o = _tensor(other, self.t)
    return TArr(fn(o, self.t) if reverse else fn(self.t, o))

  def __add__(self, o): return self._bin(o, lambda a, b: a + b)
  def __radd__(self, o): return self._bin(o, lambda a, b: a + b, True)
  def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
  def __rsub__(self, o): return self._bin(o, lambda a, b: a - b, True)
  def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
  def __rmul__(self, o): return self._bin(o, lambda a, b: a * b, True)
  def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
  def __rtruediv__(self, o): return self._bin(o, lambda a, b: a / b, True)
  def __pow__(self, o): return self._bin(o, lambda a, b: a ** b)
  def __rpow__(self, o): return self._bin(o, lambda a, b: a ** b, True)
  def __lt__(self, o): return self._bin(o, lambda a, b: a < b)
  def __le__(self, o): return self._bin(o, lambda a, b: a <= b)
  def __gt__(self, o): return self._bin(o, lambda a, b: a > b)
  def __ge__(self, o): return self._bin(o, lambda a, b: a >= b)
  def __eq__(self, o): return self._bin(o, lambda a, b: a == b)
  def __ne__(self, o): return self._bin(o, lambda a, b: a != b)
  def __and__(self, o): return self._bin(o, lambda a, b: a & b)
  def __rand__(self, o): return self._bin(o, lambda a, b: a & b, True)
  def __or__(self, o): return self._bin(o, lambda a, b: a | b)
  def __ror__(self, o): return self._bin(o, lambda a, b: a | b, True)
  def __invert__(self): return TArr(~self.t)
  def __neg__(self): return TArr(-self.t)
  def __pos__(self): return self
  def __abs__(self): return TArr(self.t.abs())
  __hash__ = None

  # -- methods numpy code calls on arrays ----------------------------------------------------------------------------
  def reshape(self, *shape):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
      shape = tuple(shape[0])
    return TArr(self.t.reshape(tuple(int(s) for s in shape)))

  def ravel(self):
    return TArr(self.t.reshape(-1))

  def copy(self):
    return TArr(self.t.clone())

  def astype(self, dtype, copy=True):
    del copy
    return TArr(self.t.to(_torch_dtype(dtype)))

  def squeeze(self, axis=None):
    return TArr(self.t.squeeze() if axis is None else self.t.squeeze(axis))

  def dot(self, other):
    o = _tensor(other, self.t)
    return TArr(self.t @ o)

  def clip(self, lo=None, hi=None):
    return TArr(self.t.clamp(lo, hi))

  def sum(self, axis=None, keepdims=False): return _reduce('sum', self, axis, keepdims)
  def mean(self, axis=None, keepdims=False): return _reduce('mean', self, axis, keepdims)
  def prod(self, axis=None, keepdims=False): return _reduce('prod', self, axis, keepdims)
  def min(self, axis=None, keepdims=False): return _reduce('amin', self, axis, keepdims)
  def max(self, axis=None, keepdims=False): return _reduce('amax', self, axis, keepdims)
  def all(self, axis=None, keepdims=False): return _reduce('all', self, axis, keepdims)
  def any(self, axis=None, keepdims=False): return _reduce('any', self, axis, keepdims)

  # -- numpy dispatch ---------------------------------------------------------------------------------------------------
  def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
    if method != '__call__' or kwargs.get('out') is not None:
      return NotImplemented
    fn = _UFUNCS.get(ufunc.__name__)
    if fn is None:
      return NotImplemented
    ref = next(x.t for x in inputs if isinstance(x, TArr))
    return TArr(fn(*[_tensor(x, ref) for x in inputs]))

  def __array_function__(self, func, types, args, kwargs):
    fn = _FUNCS.get(func)
    if fn is None:
      return NotImplemented
    return fn(*args, **kwargs)


_CONST = {}      # host constants that reached a torch op (index arrays, model scalars, targets): one device copy each


def _const(a, device, dtype=None):
  """Device tensor of the numpy array `a`, cached by content.  The task code hands the same small host arrays to every
  step (index lists from named indexing, model constants); uploading them per step would also be illegal while the step
  is being captured into a graph (a pageable host-to-device copy synchronises): the warm-up run before the capture fills
  this cache, the capture run only reads it."""
  import torch
  a = np.ascontiguousarray(a)
  k = (a.tobytes(), a.dtype.str, a.shape, str(device), dtype)
  t = _CONST.get(k)
  if t is None:
    t = torch.as_tensor(a, device=device)
    if dtype is not None and t.is_floating_point():
      t = t.to(dtype)
    _CONST[k] = t
  return t


def _torch_dtype(dtype):
  import torch
  if isinstance(dtype, torch.dtype):
    return dtype
  # numpy code says float64 for "a float": on the device that is the batch's own precision (mixing the two would also
  # stop torch's einsum / where, which do not promote)
  if np.dtype(dtype) == np.float64 and TArr.default_float is not None:
    return TArr.default_float
  return {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32,
          np.dtype(np.int64): torch.int64, np.dtype(np.bool_): torch.bool}[np.dtype(dtype)]


def _tensor(x, ref):
  """Operand of a torch op next to the tensor `ref`: TArr -> its tensor, arrays -> a tensor on ref's device, python /
  numpy floats stay scalars (torch keeps the tensor operand's dtype)."""
  import torch
  if isinstance(x, TArr):
    return x.t
  if isinstance(x, torch.Tensor):
    return x
  if isinstance(x, np.ndarray):
    if x.ndim == 0:
      return x.item()
    return _const(x, ref.device, ref.dtype if (x.dtype.kind == 'f' and ref.is_floating_point()) else None)
  if isinstance(x, (np.floating, np.integer, np.bool_)):
    return x.item()
  if isinstance(x, (list, tuple)):
    return _tensor(np.asarray(x), ref)
  return x


def _scalar(x, fl, device):
  """0-d device tensor of a python scalar, cached (see _const)."""
  import torch
  k = ('scalar', x, type(x).__name__, fl if isinstance(x, float) else None, str(device))
  t = _CONST.get(k)
  if t is None:
    t = _CONST[k] = torch.tensor(x, dtype=fl if isinstance(x, float) else None, device=device)
  return t


def _reduce(name, x, axis, keepdims):
  import torch
  t = x.t if isinstance(x, TArr) else x
  if axis is None:
    if name in ('amin', 'amax'):
      return TArr(getattr(torch, name)(t))
    return TArr(getattr(t, name)())
  axis = tuple(axis) if isinstance(axis, (tuple, list)) else int(axis)
  if name == 'prod':
    return TArr(t.prod(dim=axis, keepdim=keepdims))
  if name in ('all', 'any') and isinstance(axis, tuple):
    for a in sorted((a % t.dim() for a in axis), reverse=True):
      t = getattr(t, name)(dim=a, keepdim=keepdims)
    return TArr(t)
  return TArr(getattr(torch, name)(t, dim=axis, keepdim=keepdims))


def _first(args):
  for a in args:
    if isinstance(a, TArr):
      return a.t
    if isinstance(a, (list, tuple)):
      r = _first(a)
      if r is not None:
        return r
  return None


def _where(cond, a=None, b=None):
  import torch
  ref = _first((cond, a, b))
  c = _tensor(cond, ref)
  ta, tb = _tensor(a, ref), _tensor(b, ref)
  fl = next((t.dtype for t in (ta, tb) if isinstance(t, torch.Tensor) and t.is_floating_point()), None) or TArr.default_float or torch.float64
  if not isinstance(ta, torch.Tensor):
    ta = _scalar(ta, fl, ref.device)
  if not isinstance(tb, torch.Tensor):
    tb = _scalar(tb, fl, ref.device)
  return TArr(torch.where(c, ta, tb))


def _cat(fn):
  def run(arrays, axis=0, **kw):
    import torch
    ref = _first(arrays)
    ts = [_tensor(a, ref) for a in arrays]
    ts = [t if isinstance(t, torch.Tensor) else _const(np.asarray(t), ref.device) for t in ts]
    fl = next((t.dtype for t in ts if t.is_floating_point()), None)
    if fl is not None:
      ts = [t.to(fl) if t.is_floating_point() else t for t in ts]
    return TArr(fn(ts, dim=axis))
  return run


def _norm(x, ord=None, axis=None, keepdims=False):
  import torch
  if ord not in (None, 2):
    raise NotImplementedError('TArr: np.linalg.norm with ord=%r' % (ord,))
  return TArr(torch.linalg.vector_norm(x.t, dim=axis, keepdim=keepdims))


def _einsum(spec, *ops, **kw):
  import torch
  ref = _first(ops)
  return TArr(torch.einsum(spec, *[_tensor(o, ref) for o in ops]))


def _build_tables():
  import torch
  u = {'add': torch.add, 'subtract': torch.sub, 'multiply': torch.mul, 'true_divide': torch.div, 'divide': torch.div,
       'power': torch.pow, 'negative': torch.neg, 'absolute': torch.abs, 'fabs': torch.abs, 'sqrt': torch.sqrt,
       'square': torch.square, 'exp': torch.exp, 'log': torch.log, 'log1p': torch.log1p, 'sin': torch.sin, 'cos': torch.cos,
       'tan': torch.tan, 'arcsin': torch.arcsin, 'arccos': torch.arccos, 'arctan': torch.arctan, 'arctan2': torch.arctan2,
       'sinh': torch.sinh, 'cosh': torch.cosh, 'tanh': torch.tanh, 'arccosh': torch.arccosh, 'arctanh': torch.arctanh,
       'maximum': torch.maximum, 'minimum': torch.minimum, 'sign': torch.sign, 'floor': torch.floor, 'ceil': torch.ceil,
       'less': torch.lt, 'less_equal': torch.le, 'greater': torch.gt, 'greater_equal': torch.ge, 'equal': torch.eq,
       'not_equal': torch.ne, 'arcsinh': torch.arcsinh, 'expm1': torch.expm1, 'log2': torch.log2, 'log10': torch.log10,
       'reciprocal': torch.reciprocal, 'logical_and': torch.logical_and, 'logical_or': torch.logical_or,
       'logical_not': torch.logical_not, 'isfinite': torch.isfinite, 'isnan': torch.isnan, 'hypot': torch.hypot}

  def lift(fn):      # scalar operands next to tensors: torch's binary functions want tensors for some of them
    def run(*xs):
      ref = next(x for x in xs if isinstance(x, torch.Tensor))
      fl = ref.dtype if ref.is_floating_point() else (TArr.default_float or torch.float64)
      return fn(*[x if isinstance(x, torch.Tensor) else _scalar(x, fl, ref.device) for x in xs])
    return run
  for k in ('maximum', 'minimum', 'arctan2', 'logical_and', 'logical_or', 'hypot', 'power', 'add', 'subtract', 'multiply', 'true_divide',
            'divide', 'less', 'less_equal', 'greater', 'greater_equal', 'equal', 'not_equal'):
    u[k] = lift(u[k])
  f = {
      np.concatenate: _cat(torch.cat), np.stack: _cat(torch.stack), np.where: _where, np.linalg.norm: _norm, np.einsum: _einsum,
      np.sum: lambda x, axis=None, keepdims=False, **k: _reduce('sum', x, axis, keepdims),
      np.mean: lambda x, axis=None, keepdims=False, **k: _reduce('mean', x, axis, keepdims),
      np.prod: lambda x, axis=None, keepdims=False, **k: _reduce('prod', x, axis, keepdims),
      np.min: lambda x, axis=None, keepdims=False, **k: _reduce('amin', x, axis, keepdims),
      np.max: lambda x, axis=None, keepdims=False, **k: _reduce('amax', x, axis, keepdims),
      np.amin: lambda x, axis=None, keepdims=False, **k: _reduce('amin', x, axis, keepdims),
      np.amax: lambda x, axis=None, keepdims=False, **k: _reduce('amax', x, axis, keepdims),
      np.all: lambda x, axis=None, keepdims=False, **k: _reduce('all', x, axis, keepdims),
      np.any: lambda x, axis=None, keepdims=False, **k: _reduce('any', x, axis, keepdims),
      np.clip: lambda x, lo=None, hi=None, **k: TArr(x.t.clamp(lo, hi)),
      np.shape: lambda x: x.shape, np.ndim: lambda x: x.ndim, np.size: lambda x: x.size,
      np.reshape: lambda x, shape, **k: x.reshape(shape), np.ravel: lambda x, **k: x.ravel(),
      np.squeeze: lambda x, axis=None: x.squeeze(axis),
      np.expand_dims: lambda x, axis: TArr(x.t.unsqueeze(axis)),
      np.broadcast_to: lambda x, shape, **k: TArr(torch.broadcast_to(x.t, tuple(shape))),
      np.zeros_like: lambda x, dtype=None, **k: TArr(torch.zeros_like(x.t, dtype=_torch_dtype(dtype) if dtype else None)),
      np.ones_like: lambda x, dtype=None, **k: TArr(torch.ones_like(x.t, dtype=_torch_dtype(dtype) if dtype else None)),
      np.copy: lambda x, **k: x.copy(),
      np.dot: lambda a, b: TArr(_tensor(a, _first((a, b))) @ _tensor(b, _first((a, b)))),
      np.atleast_1d: lambda x: x if x.ndim else x.reshape(1),
      np.cross: lambda a, b, **k: TArr(torch.linalg.cross(_tensor(a, _first((a, b))), _tensor(b, _first((a, b))), dim=k.get('axis', -1))),
      np.transpose: lambda x, axes=None: TArr(x.t.permute(*axes) if axes else x.t.T),
  }
  return u, f


_UFUNCS, _FUNCS = {}, {}


def _ensure_tables():
  if not _UFUNCS:
    u, f = _build_tables()
    _UFUNCS.update(u)
    _FUNCS.update(f)


def asarray(x, dtype=None):
  """`np.asarray(x, dtype)` for the task ports: numpy arrays take exactly that call; a TArr stays on the device."""
  if isinstance(x, TArr):
    return x.astype(dtype) if dtype is not None else x
  return np.asarray(x) if dtype is None else np.asarray(x, dtype=dtype)


def array_copy(x, dtype=None):
  """`np.array(x, dtype, copy=True)` likewise."""
  if isinstance(x, TArr):
    y = x.copy()
    return y.astype(dtype) if dtype is not None else y
  return np.array(x, copy=True) if dtype is None else np.array(x, dtype=dtype, copy=True)


# ---------------------------------------------------------------------------------------------------------------------
# the view: the domain's Physics subclass over device tensors
# ---------------------------------------------------------------------------------------------------------------------
class _DevData:

  def __init__(self, view):
    object.__setattr__(self, '_v', view)

  def __getattr__(self, name):
    v = self._v
    t = v._tensors.get(name)
    if t is None:
      raise AttributeError('data.%s is not served on the device (bound fields: %s)' % (name, sorted(v._tensors)))
    return v._as_batched(name, t)

  def __setattr__(self, name, value):
    raise AttributeError('the device view is read-only: states are written through the facade at episode start')


def _make_view_class(cls):
  from dm_control_amd import physics as facade

  class DeviceView(cls):
    """`cls` (a suite domain's Physics) with `data` / `named.data` served from the batch's bound device tensors."""

    def __init__(self):      # pylint: disable=super-init-not-called
      raise TypeError('built by GenericDeviceEnv')

    def __getattr__(self, name):
      # whatever the task hung on the physics at episode start (targets ...): the host facade's value, as a device copy
      # that is refreshed IN PLACE at every restart (a captured graph keeps reading the same memory)
      if name.startswith('_'):
        raise AttributeError(name)
      host = self.__dict__['_host']
      if name in host.__dict__ or hasattr(type(host), name):
        val = getattr(host, name)
        if isinstance(val, np.ndarray) and val.ndim >= 1 and val.dtype.kind == 'f' and (val.shape[0] == self.batch_size or self.batch_size == 1):
          return self._episode_tensor(name, val)
        return val
      raise AttributeError(name)

    def _episode_tensor(self, name, val):
      import torch
      cache = self.__dict__['_episode']
      ent = cache.get(name)
      if ent is None or ent[0].shape != val.shape:
        ent = [torch.as_tensor(val, device=self._device).to(self._dtype), self._epoch]
        cache[name] = ent
      elif ent[1] != self._epoch:
        ent[0].copy_(torch.as_tensor(val, device=self._device).to(self._dtype))
        ent[1] = self._epoch
      return TArr(ent[0])

    def _as_batched(self, name, t):
      B = self.batch_size
      ncol = facade._FIELD_AXES.get(name, (None, None))[1]      # pylint: disable=protected-access
      if name in ('time', 'ncon', 'nefc', 'solver_iter'):
        return TArr(t[0])
      if ncol:
        return TArr(t.T.reshape(B, t.shape[0] // ncol, ncol))
      return TArr(t.T)

    # engine.py:589-622 accessors: copies in the reference, fresh tensors here
    def control(self): return self.data.ctrl.copy()
    def position(self): return self.data.qpos.copy()
    def velocity(self): return self.data.qvel.copy()
    def activation(self): return self.data.act.copy()
    def state(self): return self.get_state()
    def time(self): return self.data.time
    def timestep(self): return self.model.opt.timestep

    def get_state(self, sig=None):
      if sig is not None:
        raise NotImplementedError('state signatures are served by the facade')
      parts = [self.data.qpos, self.data.qvel] + ([self.data.act] if self.model.na else [])
      return np.concatenate(parts, axis=-1)

    def step(self, *a, **k): raise TypeError('the device view does not step: GenericDeviceEnv.step does')
    forward = reset = after_reset = set_control = step

    def free(self):
      pass

    def __del__(self):
      pass
  DeviceView.__name__ = 'Device' + cls.__name__
  return DeviceView


class GenericDeviceEnv:
  """B environments of `suite.<domain>.<task>` resident on one GPU; `step(action)` -> (obs (B, n), reward (B,), done
  (B,)) device tensors.  See the module docstring."""

  def __init__(self, domain, task, batch_size, precision=32, device_id=0, seed=0, capture=True, task_kwargs=None,
               termination_check_every=25, copy_outputs=True, _device='cuda'):
    import torch
    from dm_control_amd import physics as facade
    from dm_control_amd import suite
    _ensure_tables()
    self.torch = torch
    self.B = int(batch_size)
    self.device = torch.device(_device, device_id) if _device == 'cuda' else torch.device('cpu')
    self.dtype = torch.float32 if precision == 32 else torch.float64
    kw = dict(task_kwargs or {})
    kw.setdefault('random', seed)
    self.host_env = suite.load(domain, task, task_kwargs=kw, physics_kwargs=dict(batch_size=self.B, precision=precision, device_id=device_id))
    p = self.host_env.physics
    self.host_physics, self.task, self.model = p, self.host_env.task, p.model
    self.n_sub_steps = int(self.host_env._n_sub_steps)      # pylint: disable=protected-access
    lim = self.host_env._step_limit      # pylint: disable=protected-access
    self.step_limit = None if lim == float('inf') else float(lim)      # (control.Environment: done when count >= limit)
    # bind every field the facade serves from the device to a torch tensor (zero copy)
    self._tensors = {}
    names = [n for n in facade._FIELD_AXES if n not in ('xanchor', 'xaxis', 'ten_length', 'ten_velocity')]      # pylint: disable=protected-access
    for name in names + ['time', 'ncon']:
      try:
        r = p.batch._rows      # pylint: disable=protected-access
        rows, is_int = r(name) if callable(r) else (r[name] if name in r else int(np.asarray(p.batch.get(name)).shape[1]), name in ('ncon', 'nefc', 'solver_iter'))
      except Exception:      # pylint: disable=broad-except
        continue
      dt = torch.float64 if name == 'time' else torch.int32 if is_int else self.dtype
      t = torch.zeros((max(rows, 1), self.B), dtype=dt, device=self.device)
      if rows:
        cur = p.batch.get(name)
        t.copy_(torch.as_tensor(np.ascontiguousarray(np.asarray(cur).T), device=self.device).to(dt))
        if self.device.type == 'cuda':
          p.batch.bind(name, t.data_ptr())
      self._tensors[name] = t if rows else t[:0]
    View = _make_view_class(type(p))
    v = object.__new__(View)
    v.__dict__.update(_host=p, _episode={}, _epoch=0, _tensors=self._tensors, _device=self.device, _dtype=self.dtype,
                      model=p.model, batch_size=self.B)
    v.__dict__['data'] = _DevData(v)
    named = facade._Named()      # pylint: disable=protected-access
    named.model = p.named.model
    named.data = facade._Named()      # pylint: disable=protected-access
    axes = facade._make_axes(p.model)      # pylint: disable=protected-access
    for field, (rowkind, ncol) in facade._FIELD_AXES.items():      # pylint: disable=protected-access
      if field in self._tensors:
        cols = facade._Axis(facade._COLS[ncol]) if ncol else None      # pylint: disable=protected-access
        setattr(named.data, field, facade.FieldIndexer(lambda f=field: getattr(v.data, f), axes[rowkind], cols, True))
    v.__dict__['named'] = named
    self.view = v
    self.ctrl = self._tensors['ctrl']
    self.steps = 0
    self._graph = None
    self._capture = bool(capture) and self.device.type == 'cuda'
    # a replayed HIP graph writes its results into the same tensors every step: copy_outputs (default) hands the caller
    # its own copies (obs_t kept across step t + 1 stays obs_t); False returns the graph's output tensors themselves
    self._copy_outputs = bool(copy_outputs)
    self._term_every = int(termination_check_every)
    self._has_termination = type(self.task).get_termination is not _base_termination()
    # host constants that reach a torch op are cached on the device PER ENVIRONMENT (a captured graph reads them by
    # address: they must live exactly as long as this environment) and python floats take THIS environment's precision:
    # both are switched in at the start of every call (two environments of different precision in one process)
    self._consts = {}
    self._activate()
    self.reset()

  def _activate(self):
    global _CONST
    _CONST = self._consts
    TArr.default_float = self.dtype

  # -- episode start ---------------------------------------------------------------------------------------------------
  def reset(self):
    """Restarts every environment: the host port's `initialize_episode` under `reset_context` through the facade, exactly
    what `control.Environment.reset` does (rl/control.py:70-83)."""
    self._activate()
    p = self.host_physics
    p.data._invalidate()      # pylint: disable=protected-access  (the device moved since the facade last looked)
    with p.reset_context():
      self.task.initialize_episode(p)
    if self.device.type != 'cuda':
      self._pull()
    v = self.view
    v.__dict__['_epoch'] += 1
    # what the task hung on the physics (targets, radii ...): mirrored onto the view -- instance attributes, because the
    # domain classes declare them as class attributes (`target_xy = None`) that a plain lookup would find first; arrays
    # with a leading batch axis become device tensors that every restart rewrites IN PLACE
    for k, val in vars(p).items():
      if k.startswith('_') or k in ('model', 'batch', 'data', 'named', 'batch_size', 'legacy_step'):
        continue
      # (B == 1: the ports drop the batch axis -- `xy[0] if B == 1 else xy` -- so ANY float array is per-episode data that
      # must be refreshed in place: as a host array it would be baked into the captured graph at its first value)
      if isinstance(val, np.ndarray) and val.ndim >= 1 and val.dtype.kind == 'f' and (val.shape[0] == self.B or self.B == 1):
        v.__dict__[k] = v._episode_tensor(k, val)
      else:
        v.__dict__[k] = val
    self.steps = 0
    return self.observation()

  # -- task --------------------------------------------------------------------------------------------------------------
  def observation(self):
    self._activate()
    obs = self.task.get_observation(self.view)
    self.observation_layout = collections.OrderedDict((k, tuple(v.shape[1:])) for k, v in obs.items())
    ts = [v.t.reshape(self.B, -1).to(self.dtype) for v in obs.values()]
    return self.torch.cat(ts, dim=1)

  def reward(self):
    self._activate()
    r = self.task.get_reward(self.view)
    if isinstance(r, TArr):
      return r.t.to(self.dtype).reshape(self.B)
    return self.torch.full((self.B,), float(r), dtype=self.dtype, device=self.device)

  def _stream(self):
    return self.torch.cuda.current_stream().cuda_stream if self.device.type == 'cuda' else None

  def _pull(self):
    """CPU harness only (tests: torch CPU tensors next to the oracle stand-in, which has nothing to bind): the tensors
    are refreshed from the batch by copy."""
    torch = self.torch
    for name, t in self._tensors.items():
      if t.numel():
        t.copy_(torch.as_tensor(np.ascontiguousarray(np.asarray(self.host_physics.batch.get(name)).T)).to(t.dtype))

  def _control_step(self, action):
    b = self.host_physics.batch
    b.legacy_step = True
    if self.device.type != 'cuda':
      b.set('ctrl', action.numpy())
      b.step(self.n_sub_steps)
      self._pull()
    else:
      self.ctrl.copy_(action.T.to(self.dtype))
      b.step(self.n_sub_steps, stream=self._stream())
    return self.observation(), self.reward()

  def _captured_step(self, action):
    torch = self.torch
    if self._graph is None:
      getattr(self.host_physics.batch, 'wait_specialised', lambda: None)()      # (a graph keeps the kernel it was captured with)
      self._g_action = action.clone()
      state = [self._tensors[n] for n in ('qpos', 'qvel', 'qacc_warmstart', 'time', 'ctrl', 'act') if n in self._tensors]
      saved = [t.clone() for t in state]
      side = torch.cuda.Stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):              # warm-up off the default stream, as graph capture requires; it also fills
        self._control_step(self._g_action)       # the cache of device constants (_const), so that the capture run uploads nothing
      torch.cuda.current_stream().wait_stream(side)
      reads = TArr.host_reads
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph):
        self._g_out = self._control_step(self._g_action)
      if TArr.host_reads != reads:
        raise RuntimeError('the task layer read the device during capture')
      for t, v in zip(state, saved):             # the warm-up step is taken back: the replay below is this call's step
        t.copy_(v)
      self._graph = graph
    self._g_action.copy_(action)
    self._graph.replay()
    return self._g_out

  def step(self, action):
    """action: (B, nu) tensor on the device.  Returns (obs, reward, done); when the time limit is reached every
    environment restarts and the returned observation is the new episode's first.  With `capture=True,
    copy_outputs=False` obs and reward are the HIP graph's own output tensors, overwritten by the next step().
    (Per-environment episode ends and restarts without the host: suite/fused_env.py.)"""
    self._activate()
    if self._capture:
      obs, rew = self._captured_step(action)
      if self._copy_outputs:
        obs, rew = obs.clone(), rew.clone()
    else:
      obs, rew = self._control_step(action)
    self.steps += 1
    done = self.step_limit is not None and self.steps >= self.step_limit
    if not done and self._has_termination and self.steps % self._term_every == 0:
      self.host_physics.data._invalidate()      # pylint: disable=protected-access
      done = self.task.get_termination(self.host_physics) is not None
    flags = self.torch.full((self.B,), bool(done), dtype=self.torch.bool, device=self.device)
    if done:
      rew = rew.clone()
      obs = self.reset()
    return obs, rew, flags

  def warnings(self):
    return self.host_physics.batch.get('warning')

  def close(self):
    self._graph = None
    self._consts = {}
    self.host_physics.free()


def _base_termination():
  from dm_control_amd.suite import base
  return base.Task.get_termination


def make(domain, task, batch_size, **kwargs):
  return GenericDeviceEnv(domain, task, batch_size, **kwargs)
