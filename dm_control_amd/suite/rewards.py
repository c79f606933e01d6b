"""Soft indicator functions for shaping rewards (API of the reference's
dm_control/utils/rewards.py:93-139 `tolerance`), vectorised over numpy arrays so
the same task code scores a whole batch of environments at once."""
import warnings

import numpy as np

_DEFAULT_VALUE_AT_MARGIN = 0.1


def _clipped(scaled, value):
  """`value` where |scaled| < 1, else 0 (the compactly supported sigmoids)."""
  return np.where(abs(scaled) < 1, value, 0.0)


def _cosine(x, v):
  sx = x * (np.arccos(2 * v - 1) / np.pi)
  with warnings.catch_warnings():
    warnings.filterwarnings(action='ignore', message='invalid value encountered in cos')
    return _clipped(sx, (1 + np.cos(np.pi * sx)) / 2)


# name -> (f(x, v), v may be 0): f(0) = 1, f(+-1) = v, decreasing in |x|; batch-friendly numpy expressions
_SIGMOIDS = {
    'gaussian': (lambda x, v: np.exp(-0.5 * (x * np.sqrt(-2 * np.log(v)))**2), False),
    'hyperbolic': (lambda x, v: 1 / np.cosh(x * np.arccosh(1 / v)), False),
    'long_tail': (lambda x, v: 1 / ((x * np.sqrt(1 / v - 1))**2 + 1), False),
    'reciprocal': (lambda x, v: 1 / (abs(x) * (1 / v - 1) + 1), False),
    'tanh_squared': (lambda x, v: 1 - np.tanh(x * np.arctanh(np.sqrt(1 - v)))**2, False),
    'cosine': (_cosine, True),
    'linear': (lambda x, v: _clipped(x * (1 - v), 1 - x * (1 - v)), True),
    'quadratic': (lambda x, v: _clipped(x * np.sqrt(1 - v), 1 - (x * np.sqrt(1 - v))**2), True),
}


def _sigmoids(x, value_at_1, sigmoid):
  """1 at x == 0, `value_at_1` at |x| == 1, decreasing in |x|."""
  if sigmoid not in _SIGMOIDS:
    raise ValueError('Unknown sigmoid type {!r}.'.format(sigmoid))
  fn, zero_ok = _SIGMOIDS[sigmoid]
  if zero_ok:
    if not 0 <= value_at_1 < 1:
      raise ValueError('`value_at_1` must be nonnegative and smaller than 1, got {}.'.format(value_at_1))
  elif not 0 < value_at_1 < 1:
    raise ValueError('`value_at_1` must be strictly between 0 and 1, got {}.'.format(value_at_1))
  return fn(x, value_at_1)


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
