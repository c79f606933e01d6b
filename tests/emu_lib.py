"""TEST INFRASTRUCTURE: host build (LPE=1) of the kernel core, see tests/emu/emu.cpp."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'emu', 'emu.cpp')
_LIB = os.path.join(_HERE, 'emu', 'libemu.so')
_lib = None

FIELDS = ['qpos', 'qvel', 'ctrl', 'qacc_warmstart', 'qfrc_applied', 'time',
          'sensordata', 'xpos', 'xquat', 'xmat', 'xipos', 'geom_xpos', 'geom_xmat',
          'site_xpos', 'site_xmat', 'subtree_com', 'qacc', 'actuator_force',
          'qfrc_actuator', 'qfrc_bias', 'qfrc_constraint',
          'contact_dist', 'contact_pos', 'contact_frame', 'contact_force', 'cvel', 'act']
IFIELDS = ['ncon', 'nefc', 'solver_iter', 'warning', 'contact_geom1', 'contact_geom2']


def lib():
  global _lib
  if _lib is None:
    csrc = os.path.join(os.path.dirname(_HERE), 'dm_control_amd', 'csrc')
    deps = [_SRC] + [os.path.join(csrc, f) for f in ('step_core.h', 'step_layout.h', 'step_tables.h')]
    stale = lambda: not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(d) for d in deps)
    if stale():
      # (pytest-xdist workers may all find it stale at once: one builds under a lock, into a temporary that is moved in
      # place atomically -- a worker never loads a half-written object)
      import fcntl
      with open(_LIB + '.lock', 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if stale():
          tmp = _LIB + '.%d.tmp' % os.getpid()
          subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-Wno-unknown-pragmas',
                                 '-o', tmp, _SRC])
          os.replace(tmp, _LIB)
    L = ctypes.CDLL(_LIB)
    L.emu_last_error.restype = ctypes.c_char_p
    L.emu_create.restype = ctypes.c_void_p
    L.emu_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.emu_free.argtypes = [ctypes.c_void_p]
    L.emu_stash.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.emu_invalidate.argtypes = [ctypes.c_void_p]
    L.emu_set_islands.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.emu_kstash.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.emu_set_xfrc.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.emu_set_mocap.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.emu_set_env_geoms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.emu_dims.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.emu_tree_tables.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
    L.emu_find.argtypes = [ctypes.c_void_p, ctypes.c_char_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    L.emu_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    _lib = L
  return _lib


class EmuPhysics:

  def __init__(self, compiled, prec=64, nconmax=0, njmax=0, njcon=0):
    self.m = compiled
    self.prec = prec
    ints, reals = compiled.pack()
    self.h = lib().emu_create(ints.ctypes.data, ints.size, reals.ctypes.data, reals.size, nconmax, njmax, njcon)
    if not self.h:
      raise ValueError(lib().emu_last_error().decode())
    dims = np.zeros(8, dtype=np.int32)
    lib().emu_dims(self.h, dims.ctypes.data)
    self.n_sr, self.n_si, self.nconmax, self.njmax = [int(x) for x in dims[:4]]
    self.kmax = int(dims[6])
    m = compiled
    nb = m.nbody
    sizes = [m.nq, m.nv, m.nu, m.nv, m.nv, 1, m.nsensordata, 3*nb, 4*nb, 9*nb, 3*nb,
             3*m.ngeom, 9*m.ngeom, 3*m.nsite, 9*m.nsite, 3*nb, m.nv, m.nu, m.nv, m.nv, m.nv,
             self.nconmax, 3*self.nconmax, 9*self.nconmax, 6*self.nconmax, 6*nb, m.na]
    self.f = {n: np.zeros(max(s, 1)) for n, s in zip(FIELDS, sizes)}
    self._sizes = dict(zip(FIELDS, sizes))
    self.fi = {'ncon': np.zeros(1, np.int32), 'nefc': np.zeros(1, np.int32),
               'solver_iter': np.zeros(1, np.int32), 'warning': np.zeros(9, np.int32),
               'contact_geom1': np.zeros(self.nconmax, np.int32),
               'contact_geom2': np.zeros(self.nconmax, np.int32)}
    self.f['qpos'][:m.nq] = m.qpos0
    self.dbg = np.zeros(self.n_sr)
    self.dbgi = np.zeros(self.n_si, np.int32)

  def tree_tables(self):
    """StepDims::treemax models: dict(treemax, ntreetri, ntree, tree0, tree1, tri (i, j) pairs, trim)."""
    nv = self.m.nv
    dims = np.zeros(3, np.int32); t0 = np.zeros(max(nv, 1), np.int32); t1 = np.zeros(max(nv, 1), np.int32)
    tri = np.zeros(nv*(nv + 1)//2 + 1, np.int32); trim = np.zeros(nv*(nv + 1)//2 + 1, np.int32)
    lib().emu_tree_tables(self.h, dims.ctypes.data, t0.ctypes.data, t1.ctypes.data, tri.ctypes.data, trim.ctypes.data)
    n = int(dims[1])
    return dict(treemax=int(dims[0]), ntreetri=n, ntree=int(dims[2]), tree0=t0[:nv], tree1=t1[:nv],
                tri=np.stack([tri[:n] & 0xffff, tri[:n] >> 16], 1), trim=trim[:n])

  @staticmethod
  def split_solves():
    """Solves (all emulated batches of this process) whose Hessian was taken as block diagonal over the trees."""
    return int(lib().emu_split_solves_count())

  def set_islands(self, v):
    """StepOpts::islands: 1 per-island solves, 0 one joint solve, -1 by precision (fp64 on, fp32 off)."""
    lib().emu_set_islands(self.h, int(v))

  def __del__(self):
    if getattr(self, 'h', None):
      lib().emu_free(self.h)
      self.h = None

  def __getattr__(self, name):
    if name in FIELDS:
      return self.f[name][:self._sizes[name]]
    if name in IFIELDS:
      return self.fi[name]
    raise AttributeError(name)

  def _run(self, nstep, legacy, mode):
    pf = (ctypes.c_void_p * len(FIELDS))(*[self.f[n].ctypes.data for n in FIELDS])
    pi = (ctypes.c_void_p * len(IFIELDS))(*[self.fi[n].ctypes.data for n in IFIELDS])
    lib().emu_run(self.h, self.prec, pf, pi, nstep, legacy, mode, self.dbg.ctypes.data, self.dbgi.ctypes.data)

  def stash(self, on=True):
    """Keeps the position / velocity stage between legacy steps (the HBM stash of the GPU batch)."""
    lib().emu_stash(self.h, int(on))

  def kstash(self, on=True):
    """Keeps kinematics / COM frame / velocities between legacy steps (the kinematic stash of the GPU batch, on by
    default there)."""
    lib().emu_kstash(self.h, int(on))

  def invalidate(self):
    lib().emu_invalidate(self.h)

  def set_mocap(self, pos, quat):
    """mjData.mocap_pos (nmocap, 3) / mocap_quat (nmocap, 4)."""
    p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1)
    q = np.ascontiguousarray(quat, dtype=np.float64).reshape(-1)
    assert p.size == 3 * self.m.nmocap and q.size == 4 * self.m.nmocap
    lib().emu_set_mocap(self.h, p.ctypes.data, q.ctypes.data)

  def set_xfrc(self, xfrc):
    """mjData.xfrc_applied: (nbody, 6) [force, torque] at the body COMs."""
    a = np.ascontiguousarray(xfrc, dtype=np.float64).reshape(-1)
    assert a.size == 6 * self.m.nbody
    lib().emu_set_xfrc(self.h, a.ctypes.data)

  def set_env_geoms(self, geom_ids, rows):
    """Per-environment world geoms of this (single) environment: rows = (n, 16) pos / xmat / size / rbound."""
    ids = np.ascontiguousarray(geom_ids, dtype=np.int32)
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    assert rows.shape == (ids.size, 16)
    lib().emu_set_env_geoms(self.h, ids.size, ids.ctypes.data, rows.ctypes.data)

  def step(self, nstep=1, legacy=True):
    self._run(nstep, int(legacy), 0)

  def step1(self):
    self._run(1, 0, 4)

  def step2(self):
    self._run(1, 0, 5)

  def forward(self, disable_actuation=False):
    self._run(0, 0, 2 if disable_actuation else 1)

  def dense_M(self):
    import scratch_decode
    return scratch_decode.dense_M(self.m, self.scratch)

  def dense_J(self):
    import scratch_decode
    return scratch_decode.dense_J(self.m, self.scratch, self.kmax)

  def scratch(self, name):
    off, cnt, kind = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if not lib().emu_find(self.h, name.encode(), ctypes.byref(off), ctypes.byref(cnt), ctypes.byref(kind)):
      raise KeyError(name)
    src = self.dbgi if kind.value else self.dbg
    return src[off.value:off.value + cnt.value]
