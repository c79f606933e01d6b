"""dm_control_amd.mujoco_api: the `(MjModel, MjData)` seam itself (the reference's unit tests of its callers run in
tests/test_reference_mujoco.py).

  * the mjData arrays the kernel never stores, derived on the host (M, qM, qLD, xanchor / xaxis, subtree velocities,
    energy, act_dot, qfrc_passive), against the oracle's own arrays or against independent restatements;
  * mjData memory: one ndarray per field for the life of the MjData, rewritten in place; input writes uploaded;
  * model edits: in-place pushes vs batch rebuilds;
  * thread safety after dm_control/mujoco/thread_safety_test.py:53-75: independent (model, data) pairs built and stepped
    from 4 threads equal a single-threaded run bit for bit; `dmc_last_error()` is thread-local (`-m gpu`).

CPU tier: the oracle stand-in is the device.  `-m gpu`: libdmc_hip.so."""
import copy
import os
import pickle
import threading

import numpy as np
import pytest

from dm_control_amd import mjcf_compiler
from dm_control_amd import mujoco_api as mj

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dm_control_amd', 'suite', 'assets')

_BALL_CHAIN = """
<mujoco>
  <option timestep="0.002"/>
  <worldbody>
    <geom type="plane" size="5 5 .1"/>
    <body name="base" pos="0 0 1.2">
      <freejoint name="root"/>
      <geom type="box" size=".1 .06 .04" mass="2"/>
      <body name="arm" pos=".1 0 0" euler="0 20 10">
        <joint name="shoulder" type="ball" pos="0 0 0" damping=".2"/>
        <geom type="capsule" fromto="0 0 0 .3 0 0" size=".03" mass=".7"/>
        <body name="fore" pos=".3 0 0">
          <joint name="elbow_a" type="hinge" axis="0 1 0" pos="0 0 0" stiffness="3" springref=".2"/>
          <joint name="elbow_b" type="hinge" axis="0 0 1" pos="0.01 0 0"/>
          <joint name="ext" type="slide" axis="1 0 0" damping="1"/>
          <geom type="capsule" fromto="0 0 0 .25 0 0" size=".025" mass=".4"/>
          <site name="tip" pos=".25 0 0"/>
        </body>
      </body>
    </body>
  </worldbody>
  <actuator>
    <motor joint="elbow_a" gear="2"/>
    <general joint="elbow_b" dyntype="integrator" gainprm="1"/>
    <general joint="ext" dyntype="filter" dynprm=".3" gainprm="2" ctrllimited="true" ctrlrange="-1 1"/>
  </actuator>
</mujoco>
"""


@pytest.fixture
def oracle_device(monkeypatch):
  import oracle_backend as ob
  monkeypatch.setattr(mj, 'BatchedPhysics', ob.OracleBatch)
  return ob


def _model(name_or_xml):
  if name_or_xml.lstrip().startswith('<'):
    return mj.MjModel.from_xml_string(name_or_xml)
  return mj.MjModel.from_xml_path(os.path.join(_ASSETS, name_or_xml + '.xml'))


def _shake(m, d, seed=0, steps=7):
  rs = np.random.RandomState(seed)
  c = m._c
  d.qpos[:] = c.qpos0 + rs.uniform(-.3, .3, c.nq)
  for j in range(c.njnt):
    if c.jnt_type[j] in (0, 1):
      a = int(c.jnt_qposadr[j]) + (3 if c.jnt_type[j] == 0 else 0)
      q = rs.normal(size=4)
      d.qpos[a:a + 4] = q / np.linalg.norm(q)
  d.qvel[:] = rs.uniform(-1, 1, c.nv)
  d.ctrl[:] = rs.uniform(-1, 1, c.nu)
  for _ in range(steps):
    mj.mj_step(m, d)
  mj.mj_forward(m, d)


def _check_derivations(backend_oracle_of):
  for name in ('cheetah', 'humanoid', 'quadruped', _BALL_CHAIN):
    m = _model(name)
    d = mj.MjData(m)
    _shake(m, d)
    c = m._c
    nv = c.nv
    # M against the oracle's dense mass matrix (composite rigid body in the oracle, body Jacobians here)
    dense = np.zeros((nv, nv))
    mj.mju_sym2dense(dense, d.M, m.M_rownnz, m.M_rowadr, m.M_colind)
    o = backend_oracle_of(d)
    if o is not None:
      ref = np.array(o.field('qM')).reshape(nv, nv)
      np.testing.assert_allclose(dense, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()))
    full = np.zeros((nv, nv))
    mj.mj_fullM(m, full, d.qM)
    np.testing.assert_array_equal(full, dense)
    # the sparsity MuJoCo promises: M[i, j] = 0 unless one dof is an ancestor of the other
    anc = np.zeros((nv, nv), dtype=bool)
    for i in range(nv):
      j = i
      while j >= 0:
        anc[i, j] = anc[j, i] = True
        j = int(c.dof_parentid[j])
    assert np.abs(dense[~anc]).max(initial=0) < 1e-12
    # qLD: M = L' D L with L unit lower triangular on M's own pattern
    L = np.zeros((nv, nv))
    mj.mju_sym2dense(L, d.qLD, m.M_rownnz, m.M_rowadr, m.M_colind)
    D = np.diag(L).copy()
    L = np.tril(L, -1) + np.eye(nv)
    np.testing.assert_allclose(L.T @ np.diag(D) @ L, dense, rtol=0, atol=1e-10 * np.abs(dense).max())
    np.testing.assert_allclose(d.qLDiagInv, 1 / D, rtol=1e-13)
    # xanchor / xaxis against mj_kinematics' own forward walk (the facade's restatement: an independent code path)
    from dm_control_amd import physics as facade
    fake = type('P', (), {})()
    fake.batch_size, fake.model = 1, c
    fake.batch = type('B', (), {'get': staticmethod(lambda n: np.asarray(getattr(d, n), dtype=np.float64).reshape(1, -1))})()
    holder = type('D', (), {'_p': fake})()
    anchor, axis = facade._Data._joint_frames(holder)
    np.testing.assert_allclose(d.xanchor, anchor[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(d.xaxis, axis[0], rtol=0, atol=1e-12)
    # ximat = xquat * body_iquat
    for b in range(c.nbody):
      R = mjcf_compiler.quat_to_mat(mjcf_compiler.quat_mul(d.xquat[b], c.body_iquat[b]))
      np.testing.assert_allclose(d.ximat[b].reshape(3, 3), R, atol=1e-13)
    # subtree velocities against their definition, body by body
    mj.mj_subtreeVel(m, d)
    lin, angmom = d.subtree_linvel.copy(), d.subtree_angmom.copy()
    vb = np.zeros((c.nbody, 3))
    for b in range(c.nbody):
      vel = np.zeros(6)
      mj.mj_objectVelocity(m, d, int(mj.mjtObj.mjOBJ_BODY), b, vel, 0)
      vb[b] = vel[3:]
    for r in range(c.nbody):
      members = [b for b in range(c.nbody) if _in_subtree(c, b, r)]
      mass = sum(c.body_mass[b] for b in members)
      if mass <= 0:
        continue
      V = sum(c.body_mass[b] * vb[b] for b in members) / mass
      np.testing.assert_allclose(lin[r], V, atol=1e-11)
      X = d.subtree_com[r]
      Lr = np.zeros(3)
      for b in members:
        w = np.zeros(6)
        mj.mj_objectVelocity(m, d, int(mj.mjtObj.mjOBJ_BODY), b, w, 0)
        R = d.ximat[b].reshape(3, 3)
        Lr += R @ (c.body_inertia[b] * (R.T @ w[:3])) + c.body_mass[b] * np.cross(d.xipos[b] - X, vb[b] - V)
      np.testing.assert_allclose(angmom[r], Lr, atol=1e-10)


def _in_subtree(c, b, r):
  while True:
    if b == r:
      return True
    if b == 0:
      return False
    b = int(c.body_parentid[b])


def test_host_derived_mjdata_arrays_match_the_oracle_and_their_definitions(oracle_device):
  _check_derivations(lambda d: d._batch._envs[0])


@pytest.mark.gpu
def test_host_derived_mjdata_arrays_on_the_device():
  _check_derivations(lambda d: None)


def _check_subtreelinvel_sensor():
  """The cheetah's `subtreelinvel` sensor is computed by the step kernel; mjData.subtree_linvel by the host derivation."""
  m = _model('cheetah')
  d = mj.MjData(m)
  _shake(m, d, seed=3)
  mj.mj_subtreeVel(m, d)
  torso = mj.mj_name2id(m, int(mj.mjtObj.mjOBJ_BODY), 'torso')
  adr = int(m.sensor_adr[mj.mj_name2id(m, int(mj.mjtObj.mjOBJ_SENSOR), 'torso_subtreelinvel')])
  np.testing.assert_allclose(d.subtree_linvel[torso], d.sensordata[adr:adr + 3], atol=1e-11)


def test_subtree_linvel_equals_the_kernels_sensor(oracle_device):
  _check_subtreelinvel_sensor()


@pytest.mark.gpu
def test_subtree_linvel_equals_the_kernels_sensor_on_the_device():
  _check_subtreelinvel_sensor()


def _check_energy_and_actuation():
  m = _model(_BALL_CHAIN)
  m.opt.enableflags |= int(mj.mjtEnableBit.mjENBL_ENERGY)
  d = mj.MjData(m)
  _shake(m, d, seed=5)
  c = m._c
  nv = c.nv
  M = np.zeros((nv, nv))
  mj.mju_sym2dense(M, d.M, m.M_rownnz, m.M_rowadr, m.M_colind)
  assert d.energy[1] == pytest.approx(0.5 * d.qvel @ M @ d.qvel, rel=1e-13)
  pot = -sum(c.body_mass[b] * (c.opt.gravity @ d.xipos[b]) for b in range(1, c.nbody))
  a = int(c.jnt_qposadr[mj.mj_name2id(m, int(mj.mjtObj.mjOBJ_JOINT), 'elbow_a')])
  pot += 0.5 * 3 * (d.qpos[a] - c.qpos_spring[a]) ** 2
  assert d.energy[0] == pytest.approx(pot, rel=1e-12)
  # act_dot: integrator -> ctrl, filter -> (clamped ctrl - act) / tau
  d.ctrl[:] = [0.3, -0.7, 2.0]
  mj.mj_forward(m, d)
  np.testing.assert_allclose(d.act_dot, [-0.7, (1.0 - d.act[1]) / 0.3], rtol=1e-13)
  # one Euler step advances the activations by dt * act_dot
  act, dot = d.act.copy(), d.act_dot.copy()
  mj.mj_step(m, d)
  np.testing.assert_allclose(d.act, act + c.opt.timestep * dot, rtol=1e-12)
  # qfrc_passive: dampers and the hinge spring
  qp = np.zeros(nv)
  qp -= c.dof_damping * d.qvel
  qp[c.jnt_dofadr[mj.mj_name2id(m, 3, 'elbow_a')]] -= 3 * (d.qpos[a] - c.qpos_spring[a])
  np.testing.assert_allclose(d.qfrc_passive, qp, atol=1e-13)


def test_energy_activation_rates_and_passive_forces(oracle_device):
  _check_energy_and_actuation()


@pytest.mark.gpu
def test_energy_activation_rates_and_passive_forces_on_the_device():
  _check_energy_and_actuation()


def _check_memory_semantics():
  m = _model('cheetah')
  d = mj.MjData(m)
  qpos, xpos, sens, warn = d.qpos, d.xpos, d.sensordata, d.warning.number
  assert d.qpos is qpos and d.xpos is xpos
  assert not xpos.any()      # mj_makeData: nothing computed yet
  mj.mj_forward(m, d)
  assert xpos.any() and d.xpos is xpos      # rewritten in place
  z0 = xpos[1, 2]
  qpos[1] += 0.25      # root z slider: a write through the handed-out array is uploaded before the next launch
  mj.mj_forward(m, d)
  assert xpos[1, 2] == pytest.approx(z0 + 0.25, abs=1e-12)
  t0 = d.time
  mj.mj_step(m, d, 3)
  assert d.time == pytest.approx(t0 + 3 * m.opt.timestep)
  assert d.qpos is qpos and np.any(qpos != m.qpos0)
  # the state signature calls work on the same arrays
  sig = int(mj.mjtState.mjSTATE_FULLPHYSICS)
  s = np.zeros(mj.mj_stateSize(m, sig))
  mj.mj_getState(m, d, s, sig)
  assert s[0] == d.time and np.array_equal(s[1:1 + m.nq], qpos)
  s2 = s + 0.01
  mj.mj_setState(m, d, s2, sig)
  assert np.array_equal(qpos, s2[1:1 + m.nq]) and d.time == s2[0]
  # reset: inputs back to the model's, derived arrays zero, arrays still the same objects
  mj.mj_resetData(m, d)
  assert np.array_equal(qpos, m.qpos0) and not xpos.any() and d.time == 0 and not sens.any()
  # a bad control: counted by the device, visible through the SAME counter array, mjData.ctrl keeps what was written
  d.ctrl[0] = np.nan
  mj.mj_forward(m, d)
  assert warn[int(mj.mjtWarning.mjWARN_BADCTRL)] == 1 and np.isnan(d.ctrl[0])
  mj.mj_forward(m, d)
  assert warn[int(mj.mjtWarning.mjWARN_BADCTRL)] == 2
  d.warning[int(mj.mjtWarning.mjWARN_BADCTRL)].number = 0
  assert warn[int(mj.mjtWarning.mjWARN_BADCTRL)] == 0
  return m, d


def test_mjdata_arrays_are_views_that_follow_the_simulation(oracle_device):
  _check_memory_semantics()


@pytest.mark.gpu
def test_mjdata_arrays_are_views_that_follow_the_simulation_on_the_device():
  _check_memory_semantics()


def _check_model_edits():
  m = _model('cheetah')
  d = mj.MjData(m)
  batch = d._batch
  mj.mj_step(m, d, 5)
  # options and constants the batch follows in place: same batch object
  m.opt.timestep = 0.004
  m.opt.gravity[2] = -3.0
  m.dof_damping[3:] *= 2
  t = d.time
  mj.mj_step(m, d)
  assert d._batch is batch and d.time == pytest.approx(t + 0.004)
  # a reference built from scratch with those values follows the same trajectory
  m2 = _model('cheetah')
  m2.opt.timestep, m2.opt.gravity[2] = 0.004, -3.0
  m2.dof_damping[3:] *= 2
  d2 = mj.MjData(m2)
  m0 = _model('cheetah')
  d0 = mj.MjData(m0)
  mj.mj_step(m0, d0, 5)
  d2.qpos[:], d2.qvel[:], d2.qacc_warmstart[:], d2.time = d0.qpos, d0.qvel, d0.qacc_warmstart, d0.time
  mj.mj_step(m2, d2)
  np.testing.assert_allclose(d.qpos, d2.qpos, rtol=0, atol=1e-13)
  # anything else rebuilds the batch and carries the state: masses, the integrator
  m.body_mass[1:] *= 1.5
  m.opt.integrator = int(mj.mjtIntegrator.mjINT_RK4)
  q, t = d.qpos.copy(), d.time
  mj.mj_step(m, d)
  assert d._batch is not batch and d.time == pytest.approx(t + 0.004) and not np.array_equal(d.qpos, q)
  m2.body_mass[1:] *= 1.5
  m2.opt.integrator = int(mj.mjtIntegrator.mjINT_RK4)
  mj.mj_step(m2, d2)
  np.testing.assert_allclose(d.qpos, d2.qpos, rtol=0, atol=1e-12)
  # a second MjData of the same model is independent
  d3 = mj.MjData(m)
  assert d3.time == 0 and np.array_equal(d3.qpos, m.qpos0)


def test_model_edits_reach_the_device(oracle_device):
  _check_model_edits()


@pytest.mark.gpu
def test_model_edits_reach_the_device_on_the_device():
  _check_model_edits()


def _check_copies():
  m = _model('humanoid')
  d = mj.MjData(m)
  _shake(m, d, seed=2, steps=12)
  mj.mj_step(m, d)      # (xpos now belongs to the state BEFORE the integration, as in MuJoCo)
  for make in (copy.copy, copy.deepcopy, lambda x: pickle.loads(pickle.dumps(x))):
    e = make(d)
    assert e is not d and e._batch is not d._batch
    for f in ('qpos', 'qvel', 'xpos', 'sensordata', 'subtree_com', 'qacc_warmstart'):
      assert np.array_equal(getattr(e, f), getattr(d, f)), f
    assert e.time == d.time and e.ncon == d.ncon
    ref = copy.copy(d)
    for _ in range(10):
      mj.mj_step(e.model, e)
      mj.mj_step(ref.model, ref)
    assert np.array_equal(e.qpos, ref.qpos) and np.array_equal(e.xpos, ref.xpos)
  mm = pickle.loads(pickle.dumps(m))
  assert mm.nq == m.nq and np.array_equal(mm.body_pos, m.body_pos) and mm.names == m.names
  blob = np.zeros(mj.mj_sizeModel(m), dtype=np.uint8)
  mj.mj_saveModel(m, None, blob)
  m3 = mj.MjModel.from_binary_path('model.mjb', {'model.mjb': blob.tobytes()})
  assert np.array_equal(m3.geom_size, m.geom_size) and m3.opt.timestep == m.opt.timestep


def test_copy_deepcopy_pickle_continue_identically(oracle_device):
  _check_copies()


@pytest.mark.gpu
def test_copy_deepcopy_pickle_continue_identically_on_the_device():
  _check_copies()


def _check_callbacks():
  m = _model('cheetah')
  d = mj.MjData(m)
  seen = []

  def control(model, data):
    seen.append(data.time)
    data.ctrl[:] = 0.5

  def passive(model, data):
    data.qfrc_passive[0] += 7.0      # a force along the root x slider

  mj.set_mjcb_control(control)
  try:
    mj.mj_step(m, d, 3)
  finally:
    mj.set_mjcb_control(None)
  assert len(seen) == 3 and np.allclose(np.diff(seen), m.opt.timestep) and np.all(d.ctrl == 0.5)
  ref = mj.MjData(m)
  ref.ctrl[:] = 0.5
  mj.mj_step(m, ref, 3)
  assert np.array_equal(ref.qpos, d.qpos)
  mj.set_mjcb_passive(passive)
  try:
    mj.mj_step(m, d)
  finally:
    mj.set_mjcb_passive(None)
  ref.qfrc_applied[0] = 7.0
  mj.mj_step(m, ref)
  np.testing.assert_allclose(d.qpos, ref.qpos, rtol=0, atol=1e-14)
  assert not d.qfrc_applied.any()      # the callback's force was for that step only
  with pytest.raises(NotImplementedError):
    mj.set_mjcb_sensor(lambda *a: None)


def test_host_callbacks(oracle_device):
  _check_callbacks()


@pytest.mark.gpu
def test_host_callbacks_on_the_device():
  _check_callbacks()


def test_surface_the_reference_enumerates():
  """core.py builds its wrapper properties from dir(mujoco.MjModel) / dir(mujoco.MjData) and engine.py reads enum members
  at import: the names it touches are there, with MuJoCo's values where the device depends on them."""
  for n in ('nq', 'nv', 'nu', 'na', 'nbody', 'nmocap', 'nkey', 'names', 'name_bodyadr', 'body_pos', 'geom_rgba', 'jnt_qposadr',
            'jnt_dofadr', 'sensor_adr', 'actuator_actadr', 'numeric_adr', 'numeric_data', 'body_mocapid', 'opt', 'vis', 'stat',
            'M_rownnz', 'M_rowadr', 'M_colind', 'nnames', 'njmax', 'cam_pos', 'key_qpos'):
    assert n in dir(mj.MjModel), n
  for n in ('qpos', 'qvel', 'act', 'ctrl', 'time', 'xpos', 'xmat', 'xquat', 'sensordata', 'warning', 'contact', 'ncon', 'energy',
            'M', 'qLD', 'subtree_com', 'subtree_linvel', 'mocap_pos', 'mocap_quat', 'model', 'timer', 'solver', 'act_dot'):
    assert n in dir(mj.MjData), n
  assert mj.mjtIntegrator.mjINT_RK4.value == 1 and mj.mjtObj.mjOBJ_SITE == 6 and mj.mjtObj.mjOBJ_ACTUATOR == 19
  assert list(mj.mjtDisableBit.__members__)[-1] == 'mjNDISABLE' and list(mj.mjtWarning.__members__)[-1] == 'mjNWARNING'
  assert mj.mjtState.mjSTATE_FULLPHYSICS.value == (1 << 0) | (1 << 1) | (1 << 2) | (1 << 3) | (1 << 12)
  with pytest.raises(ValueError):
    mj.mjtDisableBit(-99)
  with pytest.raises(TypeError):
    mj.MjModel()
  with pytest.raises(ValueError):
    mj.MjModel.from_xml_path('/nonexistent/model.xml')
  sizes = mj.array_sizes()
  assert sizes['mjdata']['xpos'] == ('nbody', 3) and sizes['mjmodel']['geom_rgba'] == ('ngeom', 4)
  assert sizes['mjmodel']['numeric_data'] == ('nnumericdata',) and sizes['mjdata']['qLD'] == ('nC',)


def _thread_worker(out, k, steps):
  m = _model('cheetah')
  d = mj.MjData(m)
  rs = np.random.RandomState(k)
  for _ in range(steps):
    d.ctrl[:] = rs.uniform(-1, 1, m.nu)
    mj.mj_step(m, d)
  m2 = _model('hopper')
  d2 = mj.MjData(m2)      # a second, different batch in the same thread, stepped alternately
  for _ in range(steps // 2):
    mj.mj_step(m2, d2)
    mj.mj_step(m, d)
  out[k] = (d.qpos.copy(), d2.qpos.copy())


def _check_threads():
  steps, n = 40, 4
  serial, threaded = {}, {}
  for k in range(n):
    _thread_worker(serial, k, steps)
  ts = [threading.Thread(target=_thread_worker, args=(threaded, k, steps)) for k in range(n)]
  for t in ts:
    t.start()
  for t in ts:
    t.join()
  assert sorted(threaded) == list(range(n))
  for k in range(n):
    assert np.array_equal(serial[k][0], threaded[k][0]) and np.array_equal(serial[k][1], threaded[k][1]), k
  assert not np.array_equal(serial[0][0], serial[1][0])


def test_independent_model_data_pairs_step_concurrently_from_four_threads(oracle_device):
  _check_threads()


@pytest.mark.gpu
def test_independent_batches_step_concurrently_from_four_threads_on_the_device():
  """mujoco/thread_safety_test.py:53-75 on libdmc_hip.so: every thread owns its dmc_model / dmc_batch handles; launches of
  different batches interleave on the device."""
  _check_threads()


@pytest.mark.gpu
def test_dmc_last_error_is_thread_local():
  """include/dmc_batch.h: every entry point returns a status and leaves its message in a THREAD-LOCAL slot.  One thread
  provokes errors in a loop while another keeps reading its own slot, which must stay what that thread last caused."""
  import ctypes
  from dm_control_amd import _native
  L = _native.lib()
  m = _model('cheetah')
  d = mj.MjData(m)
  ptr = d._batch._ptr
  rows, is_int = ctypes.c_int(), ctypes.c_int()
  assert L.dmc_batch_field_rows(ptr, b'no_such_field_main', ctypes.byref(rows), ctypes.byref(is_int)) != 0
  mine = L.dmc_last_error()
  assert b'no_such_field_main' in mine
  stop, seen = threading.Event(), []

  def offender():
    r, i = ctypes.c_int(), ctypes.c_int()
    while not stop.is_set():
      L.dmc_batch_field_rows(ptr, b'other_thread_field', ctypes.byref(r), ctypes.byref(i))
      seen.append(L.dmc_last_error())
  t = threading.Thread(target=offender)
  t.start()
  try:
    for _ in range(2000):
      assert L.dmc_last_error() == mine
  finally:
    stop.set()
    t.join()
  assert seen and all(b'other_thread_field' in s for s in seen[:50])
  assert L.dmc_last_error() == mine
