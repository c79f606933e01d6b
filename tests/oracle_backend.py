"""TEST INFRASTRUCTURE: a stand-in for `dm_control_amd.batch.BatchedPhysics` backed by the CPU oracle.

It lets the host stack (Physics facade, named indexing, Environment, every suite task) run in the
`-m "not gpu"` tests.  It is reachable only through the `oracle_backend` pytest fixture
(tests/conftest.py), which monkeypatches the name inside `dm_control_amd.physics` for one test; the
product never imports it and still fails loudly without a GPU."""
import numpy as np

from oracle.oracle import OracleModel, OraclePhysics

_NWARN = 9
_DSBL_ACTUATION = 1 << 11


class OracleBatch:

  def __init__(self, model, batch_size, device_id=0, precision=64, nconmax=0, njmax=0, lanes_per_env=0):
    del device_id, njmax, lanes_per_env
    self.model = model
    self.batch_size = int(batch_size)
    self.precision = precision
    self.legacy_step = True
    self._om = OracleModel(model)
    self._envs = [OraclePhysics(self._om) for _ in range(self.batch_size)]
    self.nconmax = nconmax or 16
    m = model
    nb = m.nbody
    self._rows = dict(qpos=m.nq, qvel=m.nv, act=m.na, ctrl=m.nu, qacc_warmstart=m.nv, qfrc_applied=m.nv, xfrc_applied=6*nb, time=1,
                      sensordata=m.nsensordata, xpos=3*nb, xquat=4*nb, xmat=9*nb, xipos=3*nb, geom_xpos=3*m.ngeom,
                      geom_xmat=9*m.ngeom, site_xpos=3*m.nsite, site_xmat=9*m.nsite, subtree_com=3*nb, qacc=m.nv,
                      actuator_force=m.nu, qfrc_actuator=m.nv, qfrc_bias=m.nv, qfrc_constraint=m.nv, cvel=6*nb,
                      mocap_pos=3*m.nmocap, mocap_quat=4*m.nmocap)

  def close(self):
    self._envs = []

  def info(self):
    return dict(B=self.batch_size, precision=self.precision, nconmax=self.nconmax, static_id=-1)

  # -- fields ---------------------------------------------------------------------------
  def get(self, name):
    B = self.batch_size
    if name == 'warning':
      return np.stack([np.array(o.warning, dtype=np.int32) for o in self._envs])
    if name in ('ncon', 'nefc', 'solver_iter'):
      return np.array([[getattr(o, name)] for o in self._envs], dtype=np.int32)
    if name.startswith('contact_'):
      what = name[len('contact_'):]
      width = dict(dist=1, pos=3, frame=9, force=6, geom1=1, geom2=1)[what]
      out = np.zeros((B, self.nconmax * width), dtype=np.int32 if what.startswith('geom') else np.float64)
      if what.startswith('geom'):
        out[:] = -1
      for e, o in enumerate(self._envs):
        for i in range(min(o.ncon, self.nconmax)):
          c = o.contact(i)
          v = o.contact_force(i).ravel() if what == 'force' else np.ravel(c[what])
          out[e, i*width:(i + 1)*width] = v
      return out
    if name == 'time':
      return np.array([[o.time] for o in self._envs])
    if not self._rows[name]:
      return np.zeros((B, 0))
    return np.stack([np.array(o.field(name), dtype=np.float64).reshape(-1)[:self._rows[name]] for o in self._envs]).reshape(B, self._rows[name])

  def set(self, name, value):
    rows = self._rows[name]
    if not rows:
      return
    a = np.broadcast_to(np.asarray(value, dtype=np.float64).reshape((-1, rows) if np.ndim(value) > 1 else (1, rows)),
                        (self.batch_size, rows))
    for e, o in enumerate(self._envs):
      if name == 'time':
        o.time = float(a[e, 0])
      else:
        o.field(name)[:rows] = a[e]
    if name in ('qpos', 'qvel', 'act', 'mocap_pos', 'mocap_quat'):
      self._stale = True      # the HIP launch recomputes the position / velocity stage from the state it is given

  def set_control(self, control):
    self.set('ctrl', control)

  def set_opt(self, name, value):
    if name in ('disableflags', 'iterations', 'ls_iterations', 'integrator', 'cone', 'solver'):
      self._om.opt_int(name, int(value))
    else:
      self._om.opt_real(name, float(value))
    self._stale = True

  def set_model_real(self, name, values):
    self._om.field(name)[:] = np.asarray(values, dtype=np.float64).ravel()
    self._stale = True      # the device recomputes the opening stage after a model edit (stash epoch bumped)

  # -- pipeline -------------------------------------------------------------------------
  def get_many(self, names, stream=None, dtype=np.float64, copy=True):
    del stream, copy
    self.get_many_calls = getattr(self, 'get_many_calls', 0) + 1
    return {n: np.asarray(self.get(n), dtype=dtype) for n in names}

  def enable_profiling(self, enabled=True):
    self._timing = bool(enabled)

  def timer(self, which=0):
    return tuple(getattr(self, '_timers', [[0.0, 0], [0.0, 0]])[which])

  def _tick(self, which, t0, count):
    if getattr(self, '_timing', False):
      import time
      if not hasattr(self, '_timers'):
        self._timers = [[0.0, 0], [0.0, 0]]
      self._timers[which][0] += time.perf_counter() - t0
      self._timers[which][1] += count

  def step(self, nstep=1, stream=None):
    del stream
    import time
    t0 = time.perf_counter()
    self._step(nstep)
    self._tick(0, t0, int(nstep))

  def _step(self, nstep):
    for o in self._envs:
      o.legacy_step = bool(self.legacy_step)
      if getattr(self, '_stale', False) and self.legacy_step:
        o.step1()
      o.step(int(nstep))
    self._stale = False

  def step1(self, stream=None):
    del stream
    for o in self._envs:
      o.step1()
    self._stale = False

  def step2(self, stream=None):
    del stream
    for o in self._envs:
      if getattr(self, '_stale', False):
        o.step1()      # (dmc_batch_step2 recomputes the stage when the state was edited in between)
      o.step2()
    self._stale = False

  def forward(self, disable_actuation=False, stream=None):
    del stream
    import time
    t0 = time.perf_counter()
    self._forward(disable_actuation)
    self._tick(1, t0, 1)

  def _forward(self, disable_actuation):
    flags = self._om.opt_int('disableflags')
    if disable_actuation:
      self._om.opt_int('disableflags', flags | _DSBL_ACTUATION)
    for o in self._envs:
      o.forward()
    self._om.opt_int('disableflags', flags)

  def reset(self, env_mask=None, keyframe_id=None):
    from oracle.oracle import lib
    for e, o in enumerate(self._envs):
      if env_mask is None or env_mask[e]:
        lib().ora_reset(self._om.ptr, o.ptr, -1 if keyframe_id is None else int(keyframe_id))

  def sync(self):
    pass
