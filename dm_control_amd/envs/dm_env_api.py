"""Minimal `dm_env` surface (dm_env==1.6 is not installable here; SURVEY.md
Appendix D lists what the reference uses): TimeStep / StepType / restart /
transition / termination / truncation, `Environment`, `specs.Array`,
`specs.BoundedArray`.  API-compatible re-implementation, pure Python."""
import abc
import collections
import enum

import numpy as np


class StepType(enum.IntEnum):
  FIRST = 0
  MID = 1
  LAST = 2

  def first(self):
    return self is StepType.FIRST

  def mid(self):
    return self is StepType.MID

  def last(self):
    return self is StepType.LAST


class TimeStep(collections.namedtuple('TimeStep', ['step_type', 'reward', 'discount', 'observation'])):
  __slots__ = ()

  def first(self):
    return self.step_type == StepType.FIRST

  def mid(self):
    return self.step_type == StepType.MID

  def last(self):
    return self.step_type == StepType.LAST


def restart(observation):
  return TimeStep(StepType.FIRST, None, None, observation)


def transition(reward, observation, discount=1.0):
  return TimeStep(StepType.MID, reward, discount, observation)


def termination(reward, observation):
  return TimeStep(StepType.LAST, reward, 0.0, observation)


def truncation(reward, observation, discount=1.0):
  return TimeStep(StepType.LAST, reward, discount, observation)


class Environment(metaclass=abc.ABCMeta):

  @abc.abstractmethod
  def reset(self):
    """Starts a new episode, returns the first TimeStep."""

  @abc.abstractmethod
  def step(self, action):
    """Applies `action`, returns the next TimeStep."""

  @abc.abstractmethod
  def observation_spec(self):
    pass

  @abc.abstractmethod
  def action_spec(self):
    pass

  def reward_spec(self):
    return Array(shape=(), dtype=float, name='reward')

  def discount_spec(self):
    return BoundedArray(shape=(), dtype=float, minimum=0., maximum=1., name='discount')

  def close(self):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()


class Array:
  """Describes a numpy array: shape, dtype, name."""

  def __init__(self, shape, dtype, name=None):
    self._shape = tuple(int(d) for d in shape)
    self._dtype = np.dtype(dtype)
    self._name = name

  shape = property(lambda self: self._shape)
  dtype = property(lambda self: self._dtype)
  name = property(lambda self: self._name)

  def __repr__(self):
    return 'Array(shape=%r, dtype=%r, name=%r)' % (self._shape, self._dtype, self._name)

  def __eq__(self, other):
    # dm_env's specs compare shape and dtype only -- the name is a label (rl/control_test.py:93-95 relies on it)
    if not isinstance(other, Array):
      return False
    return self._shape == other._shape and self._dtype == other._dtype

  __hash__ = None

  def validate(self, value):
    value = np.asarray(value)
    if value.shape != self._shape:
      raise ValueError('Expected shape %r but found %r' % (self._shape, value.shape))
    if value.dtype != self._dtype:
      raise ValueError('Expected dtype %s but found %s' % (self._dtype, value.dtype))
    return value

  def generate_value(self):
    return np.zeros(self._shape, self._dtype)

  def replace(self, **kw):
    args = dict(shape=self._shape, dtype=self._dtype, name=self._name)
    args.update(kw)
    return type(self)(**args)


class BoundedArray(Array):

  def __init__(self, shape, dtype, minimum, maximum, name=None):
    super().__init__(shape, dtype, name)
    try:
      bmin = np.broadcast_to(minimum, shape=self.shape)
      bmax = np.broadcast_to(maximum, shape=self.shape)
    except ValueError as e:
      raise ValueError('minimum/maximum not compatible with shape: %s' % e)
    if np.any(bmin > bmax):
      raise ValueError('All values in minimum must be <= maximum')
    self._minimum = np.array(minimum, dtype=self.dtype)
    self._minimum.setflags(write=False)
    self._maximum = np.array(maximum, dtype=self.dtype)
    self._maximum.setflags(write=False)

  minimum = property(lambda self: self._minimum)
  maximum = property(lambda self: self._maximum)

  def __repr__(self):
    return 'BoundedArray(shape=%r, dtype=%r, name=%r, minimum=%s, maximum=%s)' % (
        self.shape, self.dtype, self.name, self._minimum, self._maximum)

  def __eq__(self, other):
    if not isinstance(other, BoundedArray):
      return False
    return (super().__eq__(other) and np.array_equal(self._minimum, other._minimum) and
            np.array_equal(self._maximum, other._maximum))

  __hash__ = None

  def validate(self, value):
    value = super().validate(value)
    if np.any(value < self._minimum) or np.any(value > self._maximum):
      raise ValueError('Values not in [minimum, maximum]')
    return value

  def generate_value(self):
    return (np.ones(self.shape, self.dtype) * self.dtype.type(self._minimum)).astype(self.dtype)

  def replace(self, **kw):
    args = dict(shape=self.shape, dtype=self.dtype, name=self.name,
                minimum=self._minimum, maximum=self._maximum)
    args.update(kw)
    return type(self)(**args)


class _Specs:
  Array = Array
  BoundedArray = BoundedArray


specs = _Specs()
