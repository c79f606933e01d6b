import os

import numpy as np

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets')


# Contact / constraint-row caps per domain where the library default (16 contacts) is too tight;
# the humanoid's are free: its LDS footprint puts 4 environments on a CU either way.  The
# model-specialised kernels are baked for exactly these caps (dm_control_amd/build.py).
# humanoid_CMU / the config-4 CMU model (nv = 62): four environments per CU in fp32 (tables 10 KiB + 34 .. 37 KiB of LDS
# scratch per environment; the contact rows, the sparse M and the cold tables live in global memory), so the contact
# cap is the one that never overflowed in the soak runs (48: a ragdoll lying on the floor reaches 27 .. 40 contacts;
# 32 raised mjWARN_CONTACTFULL 15 times in 0.4 M env-steps).  fp64 fits too (two environments per CU): the parity
# tests of these models run both precisions.
DEFAULT_CAPS = {'humanoid': dict(nconmax=24), 'humanoid_CMU': dict(nconmax=96),   # (64 overflowed 5 times in 300 steps of 4096 falling walkers, round 5; 96 costs no residency)
                'cmu_2019_position_floor': dict(nconmax=48),   # BASELINE config 4 physics (assets/)
                'soccer_2v2_boxhead': dict(nconmax=24),   # BASELINE config 5 physics (assets/)
                'stacker': dict(nconmax=48),   # four boxes in a heap + a folded arm: many simultaneous contacts
                'manipulator': dict(nconmax=24)}   # (the library default of 16 overflowed once in insert_peg's 300 steps x 4096: round 5)


def physics_kwargs(domain, user_kwargs):
  kw = dict(DEFAULT_CAPS.get(domain, {}))
  kw.update(user_kwargs or {})
  return kw


def read_model(filename):
  """Physics-only restatement of the reference model of the same name."""
  with open(os.path.join(_ASSETS, filename)) as f:
    return f.read()


def asarray(x, dtype=None):
  """`np.asarray(x, dtype)` for the task code: numpy inputs take exactly that call; a device array (suite/device_env.TArr:
  the same task code evaluated on GPU tensors) stays where it is."""
  if type(x).__name__ in ('TArr', 'SArr'):
    return x.astype(dtype) if dtype is not None else x
  return np.asarray(x) if dtype is None else np.asarray(x, dtype=dtype)


def array_copy(x, dtype=None):
  """`np.array(x, dtype, copy=True)` likewise."""
  if type(x).__name__ in ('TArr', 'SArr'):
    y = x.copy()
    return y.astype(dtype) if dtype is not None else y
  return np.array(x, copy=True) if dtype is None else np.array(x, dtype=dtype, copy=True)


def vnorm(x):
  """Euclidean norm over the last axis.  A single environment's vector takes the call the reference's tasks make,
  `np.linalg.norm(x)` (sqrt of a dot product): numpy's `axis=` path sums the squares in a different order and differs in
  the last bit, and the task ports are held to the reference's modules bit for bit
  (tests/test_reference_suite_domains.py)."""
  x = asarray(x)
  return np.linalg.norm(x) if x.ndim == 1 else np.linalg.norm(x, axis=-1)


def vecmat(v, mat):
  """`v . M` over the last axes: a world vector in the frame whose rotation matrix is M (`v.dot(xmat.reshape(3, 3))` in
  the reference's tasks).  A single environment takes exactly that call -- einsum accumulates in another order, a last-bit
  difference the port-vs-reference test would see."""
  v, mat = asarray(v), asarray(mat)
  if v.ndim == 1:
    return v.dot(mat.reshape(3, 3))
  return np.einsum('...i,...ij->...j', v, mat.reshape(v.shape[:-1] + (3, 3)))


def vdot(a, b):
  """Dot product over the last axis; a single environment's vectors take `np.dot` as the reference's tasks do."""
  a, b = asarray(a), asarray(b)
  return np.dot(a, b) if a.ndim == 1 else np.sum(a * b, axis=-1)
