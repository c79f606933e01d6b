"""Soft indicator functions for shaping rewards (API of the reference's
dm_control/utils/rewards.py:93-139 `tolerance`), vectorised over numpy arrays so
the same task code scores a whole batch of environments at once."""
import warnings

import numpy as np

_DEFAULT_VALUE_AT_MARGIN = 0.1


def _sigmoids(x, value_at_1, sigmoid):
  """1 at x == 0, `value_at_1` at |x| == 1, decreasing in |x|."""
  if sigmoid in ('cosine', 'linear', 'quadratic'):
    if not 0 <= value_at_1 < 1:
      raise ValueError('`value_at_1` must be nonnegative and smaller than 1, got {}.'.format(value_at_1))
  elif not 0 < value_at_1 < 1:
    raise ValueError('`value_at_1` must be strictly between 0 and 1, got {}.'.format(value_at_1))
  if sigmoid == 'gaussian':
    return np.exp(-0.5 * (x * np.sqrt(-2 * np.log(value_at_1)))**2)
  if sigmoid == 'hyperbolic':
    return 1 / np.cosh(x * np.arccosh(1 / value_at_1))
  if sigmoid == 'long_tail':
    return 1 / ((x * np.sqrt(1 / value_at_1 - 1))**2 + 1)
  if sigmoid == 'reciprocal':
    return 1 / (abs(x) * (1 / value_at_1 - 1) + 1)
  if sigmoid == 'cosine':
    sx = x * (np.arccos(2 * value_at_1 - 1) / np.pi)
    with warnings.catch_warnings():
      warnings.filterwarnings(action='ignore', message='invalid value encountered in cos')
      c = np.cos(np.pi * sx)
    return np.where(abs(sx) < 1, (1 + c) / 2, 0.0)
  if sigmoid == 'linear':
    sx = x * (1 - value_at_1)
    return np.where(abs(sx) < 1, 1 - sx, 0.0)
  if sigmoid == 'quadratic':
    sx = x * np.sqrt(1 - value_at_1)
    return np.where(abs(sx) < 1, 1 - sx**2, 0.0)
  if sigmoid == 'tanh_squared':
    return 1 - np.tanh(x * np.arctanh(np.sqrt(1 - value_at_1)))**2
  raise ValueError('Unknown sigmoid type {!r}.'.format(sigmoid))


def tolerance(x, bounds=(0.0, 0.0), margin=0.0, sigmoid='gaussian', value_at_margin=_DEFAULT_VALUE_AT_MARGIN):
  """1 inside [lower, upper]; outside, decays with the distance to the nearest
  bound measured in units of `margin` (0 if margin == 0)."""
  lower, upper = bounds
  if lower > upper:
    raise ValueError('Lower bound must be <= upper bound.')
  if margin < 0:
    raise ValueError('`margin` must be non-negative.')
  in_bounds = np.logical_and(lower <= x, x <= upper)
  if margin == 0:
    value = np.where(in_bounds, 1.0, 0.0)
  else:
    d = np.where(x < lower, lower - x, x - upper) / margin
    value = np.where(in_bounds, 1.0, _sigmoids(d, value_at_margin, sigmoid))
  return float(value) if np.isscalar(x) else value
