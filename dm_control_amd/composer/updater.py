"""Buffered / delayed / aggregated observations for a batch of composer environments on the device.

The reference gives every enabled observable an `update_interval`, a `delay`, a `buffer_size` and an `aggregator`
(dm_control/composer/observation/observable/base.py:55-63) and runs them through `observation.Updater`
(composer/observation/updater.py:120-331): `reset` takes the first sample at time 0, `prepare_for_next_control_step`
plans the samples of the coming physics steps, `update` -- called after every physics substep
(composer/environment.py:455-458) -- takes the planned ones, `get_observation` reads each observable's buffer
(composer/observation/obs_buffer.py:43-175): the last `buffer_size` samples whose arrival time `timestamp + delay` has
passed, oldest first, padded in front with zeros or with the first sample
(`ObservationPadding`, composer/environment.py:61-64), then reduced over the buffer axis by the aggregator.

For constant intervals and delays that machinery has a closed form, which is what runs here for B environments at once
with no host decision: an environment whose episode started `age` physics steps ago has taken samples k = 0, 1, 2, ...
at the times k * update_interval, and at a read the arrived ones are k < c = (age - delay) // update_interval + 1 (none
while age < delay).  Sample k lives in slot k mod R of a ring (R, B, n) -- R covers the buffer and the samples still in
flight -- and buffer row i reads sample c - buffer_size + i, or the padding when that is negative.  The reference's
planning step (`drop_unobserved_upcoming_items`) only skips samples nobody will read; it does not change what is read, and
neither does taking them (tests/test_composer_updater.py holds this class to the reference's `Updater`, value for value,
over a grid of intervals, delays, buffer sizes and control-step lengths, with environments restarting at different
times).  Episode ages are per environment, so environments that restarted at different control steps sample at their own
phases: a sample is a masked write.

What it costs: observables whose `update_interval` is a multiple of the control step (the usual case: history of past
control steps, sensor delays in control steps) are sampled once per control step and the n_sub_steps physics steps stay
ONE launch; any other interval needs the state after single substeps, so `composer.Environment` then steps one launch
per substep with an observation pass in between, as the reference does.

Not supported (raised at construction): `Variation` objects as interval / delay (the reference redraws them per sample;
the schedule then differs per environment and per draw)."""
import collections

PAD_ZERO, PAD_INITIAL = 'zero', 'initial_value'      # composer.ObservationPadding.ZERO / INITIAL_VALUE

DEFAULTS = dict(update_interval=1, buffer_size=1, delay=0, aggregator=None, corruptor=None)


def _median(torch, buf):
  """np.median over the buffer axis (axis 1 of (B, S, n)): the mean of the two middle values for an even count."""
  s = torch.sort(buf, dim=1).values
  S = buf.shape[1]
  return s[:, S // 2] if S % 2 else (s[:, S // 2 - 1] + s[:, S // 2]) / 2


AGGREGATORS = {      # observable/base.py:31-37 (`functools.partial(np_reducer, axis=0)` on one environment's (S, n) buffer)
    'min': lambda torch, buf: buf.min(dim=1).values,
    'max': lambda torch, buf: buf.max(dim=1).values,
    'mean': lambda torch, buf: buf.mean(dim=1),
    'median': _median,
    'sum': lambda torch, buf: buf.sum(dim=1),
}


class _Buffered:
  """One observable's ring: samples (R, B, n), addressed by each environment's own sample count."""

  def __init__(self, torch, name, options, B, n_sub_steps):
    self.name = name
    opt = dict(DEFAULTS)
    opt.update(options or {})
    unknown = set(opt) - set(DEFAULTS)
    if unknown:
      raise ValueError('observable %r: unknown option(s) %s' % (name, sorted(unknown)))
    for key in ('update_interval', 'buffer_size', 'delay'):
      v = opt[key] or DEFAULTS[key]      # (the reference binds a falsy attribute to its default, updater.py:79-88)
      if callable(v) or not float(v).is_integer():
        raise NotImplementedError('observable %r: %s must be a constant integer (Variations are redrawn per sample in the '
                                  'reference; the batched updater keeps one closed-form schedule)' % (name, key))
      opt[key] = int(v)
    if opt['update_interval'] < 1 or opt['buffer_size'] < 1:
      raise ValueError('observable %r: update_interval and buffer_size must be positive' % name)
    if opt['delay'] < 0:
      raise ValueError('`delay` should not be negative: got %r' % (opt['delay'],))      # obs_buffer.py:150-152
    agg = opt['aggregator']
    if agg is not None and not callable(agg):
      if agg not in AGGREGATORS:
        raise KeyError('Unrecognized aggregator name: %r. Valid names: %s.' % (agg, sorted(AGGREGATORS)))
    self.U, self.S, self.D, self.aggregator = opt['update_interval'], opt['buffer_size'], opt['delay'], agg
    # observable/base.py:129-138: the corruptor acts on every SAMPLE, before it is buffered (the aggregator sees corrupted
    # values); here a callable of the (B, ...) tensor -- its noise, if any, comes from the caller's own device generator
    self.corruptor = opt['corruptor']
    if self.corruptor is not None and not callable(self.corruptor):
      raise ValueError('observable %r: corruptor must be a callable of the (B, ...) sample' % name)
    self.ring_size = self.S + self.D // self.U + 2
    self.every_control_step = self.U % n_sub_steps == 0
    self.default = (self.U, self.S, self.D) == (1, 1, 0)      # the value at the read IS the observation: no ring
    self.ring = None
    self.first = None      # PAD_INITIAL: the episode's first sample (slot 0 holds it only until sample R overwrites it)
    self.B = B

  def _ensure(self, torch, value):
    if self.ring is None or self.ring.shape[2:] != value.shape[1:] or self.ring.dtype != value.dtype:
      self.ring = torch.zeros((self.ring_size,) + tuple(value.shape), dtype=value.dtype, device=value.device)

  def start(self, torch, value, mask, keep_first):
    """`Updater.reset` for the masked environments: sample 0 = `value` (B, ...), everything else forgotten."""
    self._ensure(torch, value)
    m = mask.reshape((self.B,) + (1,) * (value.dim() - 1))
    self.ring[0] = torch.where(m, value, self.ring[0])
    if keep_first:
      if self.first is None:
        self.first = value.clone()
      else:
        self.first.copy_(torch.where(m, value, self.first))      # (in place: carried state keeps its address under graph replay)

  def sample(self, torch, value, age, mask):
    """`Updater.update` at the environments' ages (physics steps since their `start`): those of `mask` whose age is a
    multiple of the interval store `value` as their sample age / interval."""
    if self.default:
      return
    self._ensure(torch, value)
    take = mask & (age % self.U == 0) & (age > 0)
    slot = ((age // self.U) % self.ring_size).reshape((1, self.B) + (1,) * (value.dim() - 1)).expand((1,) + tuple(value.shape))
    old = self.ring.gather(0, slot)
    m = take.reshape((1, self.B) + (1,) * (value.dim() - 1))
    self.ring.scatter_(0, slot, torch.where(m, value[None], old))

  def read(self, torch, value, age, pad, strip):
    """`Buffer.read` at the environments' ages, then the aggregator: (B, S, ...) -- (B, ...) when S == 1 and `strip`."""
    if self.default:
      buf = value[:, None]
    else:
      arrived = torch.where(age >= self.D, (age - self.D) // self.U + 1, torch.zeros_like(age))      # (B,)
      k = arrived[None, :] - self.S + torch.arange(self.S, device=age.device, dtype=age.dtype)[:, None]      # (S, B)
      have = k >= 0
      src = torch.where(have, k % self.ring_size, torch.zeros_like(k))
      shape = (self.S, self.B) + (1,) * (self.ring.dim() - 2)
      buf = self.ring.gather(0, src.reshape(shape).expand((self.S,) + tuple(self.ring.shape[1:])))
      if pad == PAD_ZERO:
        buf = torch.where(have.reshape(shape), buf, torch.zeros_like(buf))
      else:
        # (INITIAL_VALUE pads with the episode's first sample: obs_buffer.py:133-137 fills the deque with the value of the
        # first insert)
        buf = torch.where(have.reshape(shape), buf, self.first[None].expand_as(buf))
      buf = buf.transpose(0, 1)      # (B, S, ...)
    if self.aggregator is not None:
      if callable(self.aggregator):
        return self.aggregator(buf)
      return AGGREGATORS[self.aggregator](torch, buf)
    if strip and self.S == 1:
      return buf[:, 0]
    return buf


class Updater:
  """`observation.Updater` for the dict of (B, ...) tensors a batched task's `get_observation` returns.

  options: {observable name: dict(update_interval=, buffer_size=, delay=, aggregator=, corruptor=)}; names not listed keep the
  defaults (interval 1, buffer 1, no delay: the value at the end of the control step).  `pad`: PAD_ZERO / PAD_INITIAL
  (`delayed_observation_padding`), `strip_singleton_buffer_dim` as in the reference's constructor."""

  def __init__(self, torch, batch_size, n_sub_steps, options=None, pad=PAD_ZERO, strip_singleton_buffer_dim=False):
    if pad not in (PAD_ZERO, PAD_INITIAL):
      raise ValueError('pad must be %r or %r' % (PAD_ZERO, PAD_INITIAL))
    self.torch, self.B, self.n = torch, int(batch_size), int(n_sub_steps)
    self._options = dict(options or {})
    self._pad, self._strip = pad, bool(strip_singleton_buffer_dim)
    self._buffers = collections.OrderedDict()
    self._made = set()
    for name, opt in self._options.items():
      self._buffers[name] = _Buffered(torch, name, opt, self.B, self.n)
    self.age = None      # (B,) physics steps since the environment's episode began

  @property
  def needs_substeps(self):
    """True when some observable must be sampled between the physics steps of a control step."""
    return any(not (b.default or b.every_control_step) for b in self._buffers.values())

  def _buffer(self, name):
    b = self._buffers.get(name)
    if b is None:
      b = self._buffers[name] = _Buffered(self.torch, name, None, self.B, self.n)
    return b

  def _check(self, obs):
    missing = [n for n in self._options if n not in obs]
    if missing:
      raise KeyError('observation options name unknown observable(s): %s' % missing)

  def corrupt(self, obs):
    """The observables' corruptors on one pass of raw values (call ONCE per pass: `sample`, `start` and `read` of a pass must see
    the same corrupted sample)."""
    if not any(b.corruptor is not None for b in self._buffers.values()):
      return obs
    out = type(obs)()
    for name, value in obs.items():
      b = self._buffers.get(name)
      out[name] = b.corruptor(value) if b is not None and b.corruptor is not None else value
    return out

  def start(self, obs, mask):
    """Episode start of the masked environments with their first observation `obs` (dict of (B, ...))."""
    torch = self.torch
    self._check(obs)
    if self.age is None:
      self.age = torch.zeros(self.B, dtype=torch.int64, device=mask.device)
    self.age.masked_fill_(mask, 0)
    for name, value in obs.items():
      b = self._buffer(name)
      if b.default:
        continue
      b.start(torch, value, mask, self._pad == PAD_INITIAL)

  def advance(self, steps, mask):
    """`steps` physics steps have passed for the masked environments."""
    self.age.add_(mask.to(self.age.dtype) * int(steps))

  def sample(self, obs, mask):
    """The observation pass after a physics step (`Updater.update`) for the masked environments, at their current ages."""
    for name, value in obs.items():
      self._buffer(name).sample(self.torch, value, self.age, mask)

  def read(self, obs):
    """`Updater.get_observation` at the end of a control step: `obs` are the current values (what the default observables return)."""
    out = type(obs)()
    for name, value in obs.items():
      out[name] = self._buffer(name).read(self.torch, value, self.age, self._pad, self._strip)
    return out
