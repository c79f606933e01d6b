"""Parity tests proper: the HIP kernel, called through the C-ABI, against the fp64
CPU oracle on identical seeded inputs.  Tolerances (north_star): rel qpos error
<= 1e-4 over 1000 steps for the fp32 production kernel; the fp64 kernel must
track the oracle to 1e-9."""
import os

import numpy as np
import scratch_decode
import pytest

from dm_control_amd import mjcf_compiler as mc

pytestmark = pytest.mark.gpu

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      'dm_control_amd', 'suite', 'assets')
# north_star: "within 1e-4 rel qpos error vs CPU mj_step over 1000 steps", to a
# stated fp64/fp32 tolerance.  The fp64 kernel holds 1e-9 for every environment.
# The fp32 kernel seeds ~1e-7 rounding differences which contact events amplify
# (the reference itself demonstrates this sensitivity,
# dm_control/mujoco/tutorial.ipynb:1122-1160).  Measured over 256 environments x 1000 steps
# (profiles/r02_parity_dist_cheetah_f32.json): median 1.6e-6, 95 % <= 4.6e-6, 98.8 % <= 1e-4; the three
# environments above 1e-4 stay below it for 650+ steps and then diverge through a contact that
# closes one step earlier or later (their per-step, teacher-forced error never exceeds 1e-5).  Held to:
# (Round 3, relative line-search cost in fp32, 512 environments x 1000 steps, profiles/r03_parity_dist_cheetah_f32_lanes{16,32,64}.json:
# 99.4 % <= 1e-4 at 16 / 32 lanes, 98.4 % at 64 lanes -- a different summation order, other environments in the tail;
# teacher-forced per-step error median 3e-7, 99 % <= 9.5e-7 for all three.)
# median <= 5e-6, at least 98 % of environments <= 1e-4 on the production shape (north_star), and every environment above 1e-4 is justified
# individually: replayed teacher-forced (the kernel restarted from the oracle's state at every one of its 1000
# steps, same actions) its per-step error must stay below 5e-5 -- the open-loop gap is then amplification of
# rounding by the dynamics, not a defect of the step (stiff contacts: |qacc| ~ 1e3..1e4 with ~1e-5 relative
# error, times dt^2) -- except on the rare steps where a contact is just touching (|dist| < 1e-6) and fp32 / fp64
# disagree about activating it, which the replay identifies and reports (_teacher_forced_replay).
TOL_F64_1000 = 1e-9
TOL_F32_MEDIAN, TOL_F32_FRAC_1E4 = 5e-6, 0.98      # production shape (32 lanes, 256 envs); the 64-env shapes: at most 5 environments
TOL_F32_ONE_STEP = 5e-5


def _model(name):
  with open(os.path.join(ASSETS, name + '.xml')) as f:
    return mc.compile_xml(f.read())


@pytest.fixture(scope='module')
def cheetah():
  return _model('cheetah')


def _batch(model, B, **kw):
  from dm_control_amd.batch import BatchedPhysics
  return BatchedPhysics(model, B, **kw)


def _oracles(model, q, v=None):
  from oracle.oracle import OraclePhysics
  out = []
  for e in range(q.shape[0]):
    p = OraclePhysics(model)
    p.qpos[:] = q[e]
    if v is not None:
      p.qvel[:] = v[e]
    p.forward()
    out.append(p)
  return out


def _rel_err_env(qg, qo):
  return np.abs(qg - qo).max(axis=1) / np.maximum(1.0, np.abs(qo).max(axis=1))


def _rel_err(qg, qo):
  return float(_rel_err_env(qg, qo).max())


def _cheetah_init(model, n, seed0=0):
  # Cheetah.initialize_episode (suite/cheetah.py:63-76): limited joints ~ U(range)
  q = np.tile(model.qpos0, (n, 1))
  lim = model.jnt_limited == 1
  lo, hi = model.jnt_range[lim].T
  for e in range(n):
    q[e, lim] = np.random.RandomState(seed0 + e).uniform(lo, hi)
  return q


@pytest.mark.parametrize('lanes', [64, 32, 16])
def test_forward_stages_fp64(cheetah, lanes):
  m = cheetah
  NE = 16
  rs = np.random.RandomState(0)
  q = np.tile(m.qpos0, (NE, 1))
  q[:, 3:] += rs.uniform(-0.4, 0.4, (NE, 6))
  q[:, 2] += rs.uniform(-0.5, 0.5, NE)
  q[:, 1] = rs.uniform(-0.6, 0.1, NE)
  v = rs.uniform(-1, 1, (NE, m.nv))
  c = rs.uniform(-1, 1, (NE, m.nu))
  b = _batch(m, NE, precision=64, lanes_per_env=lanes)
  b.debug_enable(NE)
  b.set('qpos', q); b.set('qvel', v); b.set('ctrl', c)
  b.forward()
  from oracle.oracle import OraclePhysics
  for e in range(NE):
    o = OraclePhysics(m)
    o.qpos[:], o.qvel[:], o.ctrl[:] = q[e], v[e], c[e]
    o.forward()
    im = b.debug_get('imisc', e)
    assert int(im[0]) == o.ncon and int(im[1]) == o.nefc
    ne = o.nefc
    for name, ref in (('xpos', o.xpos), ('xmat', o.xmat), ('cdof', o.cdof),
                      ('qfrc_bias', o.qfrc_bias),
                      ('efc_D', o.efc_D[:ne]), ('efc_aref', o.efc_aref[:ne])):
      np.testing.assert_allclose(b.debug_get(name, e)[:ref.size], ref, rtol=1e-9, atol=1e-11,
                                 err_msg='%s env %d' % (name, e))
    # the kernel's sparse M and class-compressed Jacobian, expanded to the oracle's dense form
    get = lambda n, e=e: b.debug_get(n, e)
    np.testing.assert_allclose(scratch_decode.dense_M(m, get), o.qM, rtol=1e-9, atol=1e-11, err_msg='qM env %d' % e)
    np.testing.assert_allclose(scratch_decode.dense_J(m, get, b.info()['jac_kmax']), o.efc_J[:ne*m.nv], rtol=1e-9, atol=1e-11,
                               err_msg='efc_J env %d' % e)
    np.testing.assert_allclose(b.debug_get('qacc', e), o.qacc, rtol=1e-10, atol=1e-8)
  # derived outputs through the public field API
  np.testing.assert_allclose(b.get('ncon')[:, 0], [int(b.debug_get('imisc', e)[0]) for e in range(NE)])
  b.close()


def _teacher_forced_replay(m, q0, acts, lanes):
  """Per-env max over the steps of the fp32 kernel's one-step error when it is restarted from the oracle's state at
  every step of the same episode (q0: (n, nq) initial configurations, acts: (T, n, nu)).

  Steps on which the two DISAGREE ABOUT A CONTACT THAT IS JUST TOUCHING are reported separately: MuJoCo activates a
  contact when dist < margin, and an activated contact pushes back at once through the damping term of solref, so
  the dynamics are discontinuous there; with |dist - margin| below fp32 resolution (positions ~1 m: 1e-7) the last
  bit decides, and a one-step gap of ~1e-2 follows (measured: cheetah's back foot grazing the ground at dist =
  -6e-8 in fp32, >= 0 in fp64).  Returns (per-env max over the other steps, list of (env, step, dist) events)."""
  from oracle import oracle
  n = len(q0)
  refs = _oracles(m, q0)
  oracle.rollout_legacy(refs, np.zeros((200, n, m.nu)))
  b = _batch(m, n, precision=32, lanes_per_env=lanes)
  worst = np.zeros(n)
  events = []
  for t in range(acts.shape[0]):
    b.set('qpos', np.stack([p.qpos for p in refs]))
    b.set('qvel', np.stack([p.qvel for p in refs]))
    b.set('qacc_warmstart', np.stack([p.qacc_warmstart for p in refs]))
    b.set_control(acts[t])
    b.forward()          # the contact set the kernel sees at the oracle's state
    ncon, dist = b.get('ncon')[:, 0], b.get('contact_dist')
    g1, g2 = b.get('contact_geom1'), b.get('contact_geom2')
    edge = np.zeros(n, bool)
    for e, p in enumerate(refs):
      mine = {(int(g1[e, c]), int(g2[e, c])): float(dist[e, c]) for c in range(int(ncon[e]))}
      theirs = {}
      for c in range(p.ncon):
        cc = p.contact(c)
        theirs[(cc['geom1'], cc['geom2'])] = cc['dist']
      only = [(k, d) for k, d in mine.items() if k not in theirs] + [(k, d) for k, d in theirs.items() if k not in mine]
      if only:
        assert all(abs(d) < 1e-6 for _, d in only), ('contact sets differ beyond rounding', e, t, only)   # margin = 0
        edge[e] = True
        events.append((e, t, only[0][1]))
    b.set('qacc_warmstart', np.stack([p.qacc_warmstart for p in refs]))      # mj_forward left its own solution there
    b.step()
    oracle.rollout_legacy(refs, acts[t:t + 1])
    err = _rel_err_env(b.get('qpos'), np.stack([p.qpos for p in refs]))
    worst = np.maximum(worst, np.where(edge, 0.0, err))
  b.close()
  return worst, events


@pytest.mark.parametrize('precision,lanes,NE', [(64, 64, 32), (64, 16, 32), (32, 64, 64), (32, 32, 256), (32, 16, 64)])
def test_cheetah_1000_step_rollout(cheetah, precision, lanes, NE):
  """BASELINE config 2 on a subset of environments (256 for the production shape): task initialisation (random
  limited joints + 200 settle steps), then 1000 random-action steps, open loop."""
  from oracle import oracle
  m = cheetah
  T = 1000
  q = _cheetah_init(m, NE)
  b = _batch(m, NE, precision=precision, lanes_per_env=lanes)
  b.set('qpos', q)
  refs = _oracles(m, q)
  b.step(200)
  oracle.rollout_legacy(refs, np.zeros((200, NE, m.nu)))
  b.set('time', np.zeros((NE, 1)))
  rs = np.random.RandomState(0)
  acts = rs.uniform(-1, 1, (T, NE, m.nu)).astype(np.float32).astype(np.float64)
  worst = _rel_err_env(b.get('qpos'), np.stack([p.qpos for p in refs]))
  for t in range(T):
    b.set_control(acts[t])
    b.step()
    oracle.rollout_legacy(refs, acts[t:t + 1])
    if t % 10 == 9 or t == T - 1:
      worst = np.maximum(worst, _rel_err_env(b.get('qpos'), np.stack([p.qpos for p in refs])))
  print('precision %d lanes %d: per-env max rel qpos err over %d steps: median %.2e max %.2e' %
        (precision, lanes, T, np.median(worst), worst.max()))
  if precision == 64:
    assert worst.max() < TOL_F64_1000, worst.max()
  else:
    assert np.median(worst) < TOL_F32_MEDIAN, np.median(worst)
    assert np.sum(worst > 1e-4) <= (5 if NE < 256 else int(round((1 - TOL_F32_FRAC_1E4) * NE))), np.sort(worst)[-8:]
    tail = np.nonzero(worst > 1e-4)[0]
    if tail.size:
      step_err, events = _teacher_forced_replay(m, q[tail], acts[:, tail], lanes)
      print('  environments above 1e-4: %s open-loop %s, teacher-forced per-step max %s; just-touching contacts decided '
            'by the last bit (env, step, dist): %s'
            % (tail.tolist(), np.array2string(worst[tail], precision=2), np.array2string(step_err, precision=2),
               [(int(tail[e]), t, float('%.2g' % d)) for e, t, d in events]))
      assert step_err.max() < TOL_F32_ONE_STEP, (tail, step_err)
      assert len(events) <= 2 * tail.size, events
  assert not b.get('warning').any()
  ok = worst < 1e-4
  sens = b.get('sensordata')
  np.testing.assert_allclose(sens[ok], np.stack([p.sensordata for p in refs])[ok],
                             atol=1e-7 if precision == 64 else 1e-2)
  np.testing.assert_allclose(b.get('time')[:, 0], T * m.opt.timestep, rtol=1e-5)
  b.close()


@pytest.mark.parametrize('lanes', [64, 16])
def test_teacher_forced_single_step_fp32(cheetah, lanes):
  """fp32 kernel restarted from the oracle's state every step: per-step error."""
  from oracle import oracle
  m = cheetah
  NE, T = 32, 200
  q = _cheetah_init(m, NE, seed0=50)
  refs = _oracles(m, q)
  oracle.rollout_legacy(refs, np.zeros((200, NE, m.nu)))
  b = _batch(m, NE, precision=32, lanes_per_env=lanes)
  rs = np.random.RandomState(9)
  worst = worst_dv = 0.0
  for t in range(T):
    a = rs.uniform(-1, 1, (NE, m.nu))
    b.set('qpos', np.stack([p.qpos for p in refs]))
    b.set('qvel', np.stack([p.qvel for p in refs]))
    b.set('qacc_warmstart', np.stack([p.qacc_warmstart for p in refs]))
    b.set_control(a)
    b.step()
    oracle.rollout_legacy(refs, a[None])
    worst = max(worst, _rel_err(b.get('qpos'), np.stack([p.qpos for p in refs])))
    dv = np.abs(b.get('qvel') - np.stack([p.qvel for p in refs])).max()
    worst_dv = max(worst_dv, dv)
    assert dv < 2e-3, dv      # |qvel| ~ 10..20 through stiff contacts: measured 8e-5 .. 1.3e-3 depending on the summation order of J'DJ
  print('measured: teacher_forced_single_step_fp32 max rel dqpos=%.3g max|dqvel|=%.3g' % (worst, worst_dv))
  assert worst < TOL_F32_ONE_STEP, worst
  b.close()


def test_solver_iterations_bounded_in_fp32(cheetah):
  """The fp32 solver must stop at the rounding floor instead of running to
  opt.iterations (a tail that would stall every launch)."""
  m = cheetah
  B = 1024
  b = _batch(m, B, precision=32)
  b.set('qpos', _cheetah_init(m, B))
  b.step(200)
  rs = np.random.RandomState(4)
  worst = 0
  for _ in range(100):
    b.set_control(rs.uniform(-1, 1, (B, m.nu)))
    b.step()
    worst = max(worst, int(b.get('solver_iter').max()))
  assert worst <= 20, worst
  b.close()


def test_fused_nstep_equals_single_steps(cheetah):
  m = cheetah
  q = _cheetah_init(m, 8)
  a, b = _batch(m, 8, precision=32), _batch(m, 8, precision=32)
  for x in (a, b):
    x.set('qpos', q)
    x.set_control(np.full((8, m.nu), 0.25))
  for _ in range(6):
    a.step(1)
  b.step(6)
  np.testing.assert_array_equal(a.get('qpos'), b.get('qpos'))
  np.testing.assert_array_equal(a.get('qvel'), b.get('qvel'))
  np.testing.assert_array_equal(a.get('sensordata'), b.get('sensordata'))
  a.close(); b.close()


def test_legacy_vs_nonlegacy_outputs(cheetah):
  # legacy_step: derived fields belong to the NEW state (trailing mj_step1,
  # engine.py:147-162); non-legacy: to the state before the last integration.
  m = cheetah
  q = _cheetah_init(m, 4)
  a, b = _batch(m, 4, precision=64), _batch(m, 4, precision=64)
  for x in (a, b):
    x.set('qpos', q)
  b.legacy_step = False
  a.step(3); b.step(3)
  np.testing.assert_array_equal(a.get('qpos'), b.get('qpos'))
  refs = _oracles(m, a.get('qpos'), a.get('qvel'))
  np.testing.assert_allclose(a.get('xpos'), np.stack([p.xpos for p in refs]), atol=1e-12)
  assert np.abs(a.get('xpos') - b.get('xpos')).max() > 1e-6
  a.close(); b.close()


@pytest.mark.parametrize('B', [1, 3, 5, 67])
def test_ragged_batch_sizes_and_slot_independence(cheetah, B):
  """An environment's result must not depend on which slot / workgroup it sits in."""
  m = cheetah
  q1 = _cheetah_init(m, 1, seed0=7)
  ref = _batch(m, 1, precision=32)
  ref.set('qpos', q1)
  ref.step(25)
  want = ref.get('qpos')[0]
  b = _batch(m, B, precision=32)
  q = _cheetah_init(m, B, seed0=100)
  q[B - 1] = q1[0]
  b.set('qpos', q)
  b.step(25)
  np.testing.assert_array_equal(b.get('qpos')[B - 1], want)
  assert np.all(np.isfinite(b.get('qpos')))
  ref.close(); b.close()


def test_full_batch_4096_properties(cheetah):
  """BASELINE size: determinism, permutation equivariance, agreement of a sparse
  sample with the oracle, no warnings."""
  from oracle import oracle
  m = cheetah
  B, T = 4096, 50
  q = _cheetah_init(m, B)
  rs = np.random.RandomState(3)
  acts = rs.uniform(-1, 1, (T, B, m.nu))
  perm = rs.permutation(B)

  def run(qq, aa):
    b = _batch(m, B, precision=32)
    b.set('qpos', qq)
    for t in range(T):
      b.set_control(aa[t])
      b.step()
    out = b.get('qpos'), b.get('sensordata'), b.get('warning')
    b.close()
    return out
  q1, s1, w1 = run(q, acts)
  q2, s2, _ = run(q, acts)
  np.testing.assert_array_equal(q1, q2)
  np.testing.assert_array_equal(s1, s2)
  q3, _, _ = run(q[perm], acts[:, perm])
  np.testing.assert_array_equal(q3, q1[perm])
  assert not w1.any()
  sample = np.arange(0, B, 257)
  refs = _oracles(m, q[sample])
  oracle.rollout_legacy(refs, acts[:, sample])
  assert _rel_err(q1[sample], np.stack([p.qpos for p in refs])) < 1e-4


def test_reset_mask_and_keyframe():
  m = mc.compile_xml("""
  <mujoco><worldbody><body><joint name="a" type="hinge" axis="0 1 0"/><geom size=".1" pos=".3 0 0"/>
  </body></worldbody><keyframe><key qpos="0.5" qvel="-1"/></keyframe></mujoco>""")
  b = _batch(m, 4, precision=64)
  b.step(10)
  before = b.get('qpos').copy()
  b.reset(env_mask=[1, 0, 0, 1])
  after = b.get('qpos')
  assert after[0, 0] == 0 and after[3, 0] == 0
  np.testing.assert_array_equal(after[1:3], before[1:3])
  assert b.get('time')[0, 0] == 0 and b.get('time')[1, 0] > 0
  b.reset(keyframe_id=0)
  np.testing.assert_array_equal(b.get('qpos'), np.full((4, 1), 0.5))
  np.testing.assert_array_equal(b.get('qvel'), np.full((4, 1), -1.0))
  b.close()


def test_bad_state_raises_warning_and_resets_only_that_env(cheetah):
  # engine_test.py:502-523 semantics, per environment
  m = cheetah
  W = mc.C
  b = _batch(m, 4, precision=32)
  q = np.tile(m.qpos0, (4, 1))
  q[1, 0] = np.inf
  q[2, 3] = np.nan
  b.set('qpos', q)
  c = np.zeros((4, m.nu))
  c[3, 0] = np.nan
  b.set_control(c)
  b.step()
  w = b.get('warning')
  assert w[1, W['DMC_WARN_BADQPOS']] == 1 and w[2, W['DMC_WARN_BADQPOS']] == 1
  assert w[3, W['DMC_WARN_BADCTRL']] == 1
  assert not w[0].any()
  assert np.all(np.isfinite(b.get('qpos')))
  b.close()


def test_contact_cap_warning(cheetah):
  m = cheetah
  b = _batch(m, 2, precision=32, nconmax=2, njmax=10)
  q = np.tile(m.qpos0, (2, 1))
  q[1, 1] = -0.62
  b.set('qpos', q)
  b.forward()
  w = b.get('warning')
  assert w[1, mc.C['DMC_WARN_CONTACTFULL']] >= 1 and not w[0].any()
  assert b.get('ncon')[1, 0] == 2
  b.close()


def test_readme_golden_through_the_hip_path():
  # dm_control/mujoco/README.md:10-49 on the GPU kernel (fp64): plane-box contacts,
  # slide joint, pyramidal cone, Newton, Euler.
  m = mc.compile_xml("""
  <mujoco><worldbody>
    <geom name="floor" type="plane" size="1 1 .1"/>
    <body name="box" pos="0 0 .3">
      <joint name="up_down" type="slide" axis="0 0 1"/>
      <geom name="box" type="box" size=".2 .2 .2"/>
      <geom name="sphere" pos=".2 .2 .2" size=".1"/>
    </body></worldbody></mujoco>""")
  b = _batch(m, 2, precision=64)
  b.set('qpos', np.full((2, 1), 0.5))
  b.forward(disable_actuation=True)
  np.testing.assert_allclose(b.get('geom_xpos')[0].reshape(-1, 3), [[0, 0, 0], [0, 0, .8], [.2, .2, 1.]], atol=1e-12)
  while b.get('time')[0, 0] < 1.:
    b.step(10)
  # 1.0 s is not an exact multiple in floating point: replicate `while time < 1: step()`
  b2 = _batch(m, 1, precision=64)
  b2.set('qpos', np.full((1, 1), 0.5))
  n = 0
  t = 0.0
  while t < 1.:
    t += m.opt.timestep
    n += 1
  b2.step(n)
  z = b2.get('geom_xpos')[0].reshape(-1, 3)[1:, 2]
  np.testing.assert_allclose(z, [0.19996362, 0.39996362], atol=5e-9)
  b.close(); b2.close()


def test_zero_copy_device_binding(cheetah):
  import torch
  m = cheetah
  B = 16
  b = _batch(m, B, precision=32)
  ctrl = torch.full((m.nu, B), 0.5, dtype=torch.float32, device='cuda')
  b.bind('ctrl', ctrl.data_ptr())
  b.step(5, stream=torch.cuda.current_stream().cuda_stream)
  torch.cuda.synchronize()
  a = _batch(m, B, precision=32)
  a.set_control(np.full((B, m.nu), 0.5))
  a.step(5)
  np.testing.assert_array_equal(a.get('qpos'), b.get('qpos'))
  a.close(); b.close()


@pytest.mark.parametrize('precision,nsub', [(32, 1), (64, 3)])
def test_rollout_equals_per_step_loop(cheetah, precision, nsub):
  """One-launch rollout (per-step controls and outputs in (T, rows, B) device
  buffers) must reproduce the launch-per-step loop bit for bit."""
  import torch
  m = cheetah
  B, T = 16, 25
  td = torch.float32 if precision == 32 else torch.float64
  q = _cheetah_init(m, B, seed0=20)
  rs = np.random.RandomState(1)
  acts = rs.uniform(-1, 1, (T, B, m.nu))
  loop = _batch(m, B, precision=precision)
  loop.set('qpos', q)
  want_q, want_v, want_s = [], [], []
  for t in range(T):
    loop.set_control(acts[t])
    loop.step(nsub)
    want_q.append(loop.get('qpos')); want_v.append(loop.get('qvel')); want_s.append(loop.get('sensordata'))
  ro = _batch(m, B, precision=precision)
  ro.set('qpos', q)
  ctrl = torch.from_numpy(np.ascontiguousarray(acts.transpose(0, 2, 1))).to('cuda').to(td).contiguous()
  qs = torch.zeros((T, m.nq, B), dtype=td, device='cuda')
  vs = torch.zeros((T, m.nv, B), dtype=td, device='cuda')
  ss = torch.zeros((T, m.nsensordata, B), dtype=td, device='cuda')
  ro.rollout(T, nsub, ctrl.data_ptr(), qs.data_ptr(), vs.data_ptr(), ss.data_ptr())
  ro.sync()
  np.testing.assert_array_equal(qs.cpu().numpy().transpose(0, 2, 1).astype(np.float64), np.stack(want_q))
  np.testing.assert_array_equal(vs.cpu().numpy().transpose(0, 2, 1).astype(np.float64), np.stack(want_v))
  np.testing.assert_array_equal(ss.cpu().numpy().transpose(0, 2, 1).astype(np.float64), np.stack(want_s))
  np.testing.assert_array_equal(ro.get('qpos'), loop.get('qpos'))
  np.testing.assert_array_equal(ro.get('sensordata'), loop.get('sensordata'))
  np.testing.assert_allclose(ro.get('time'), loop.get('time'), rtol=0, atol=0)
  loop.close(); ro.close()


# ---- elliptic friction cones (cone="elliptic": suite finger / stacker / manipulator) ---------------
_ELLIPTIC_SCENE = """
<mujoco><option cone="elliptic" impratio="{impratio}" gravity="2 0.5 -9.81"/>
<default><geom friction="0.7 0.02 0.003" condim="{condim}"/></default>
<worldbody>
  <geom name='floor' type='plane' size='5 5 1' conaffinity='5'/>
  <body name='box' pos='0 0 .12'><freejoint/>
    <geom name='box' type='box' size='.1 .08 .1' contype='4' conaffinity='4'/>
    <body name='arm' pos='.1 0 .1'><joint name='h' type='hinge' axis='0 1 0' range='-60 60' limited='true'/>
      <geom name='arm' type='capsule' fromto='0 0 0 .3 0 0' size='.04'/></body></body>
  <body name='ball' pos='.5 .3 .1'><freejoint/><geom name='ball' size='.1'/></body>
</worldbody></mujoco>"""


@pytest.mark.parametrize('precision,condim,impratio,lanes', [(64, 3, 1.0, 64), (64, 4, 5.0, 32), (64, 6, 1.0, 16),
                                                           (32, 3, 1.0, 0), (32, 6, 2.0, 0)])
def test_elliptic_cones_match_oracle(precision, condim, impratio, lanes):
  m = mc.compile_xml(_ELLIPTIC_SCENE.format(condim=condim, impratio=impratio))
  B = 24
  rs = np.random.RandomState(11)
  q = np.tile(m.qpos0, (B, 1))
  v = rs.uniform(-1, 1, (B, m.nv))
  b = _batch(m, B, precision=precision, lanes_per_env=lanes)
  b.set('qpos', q); b.set('qvel', v)
  ora = _oracles(m, q, v)
  b.forward()
  np.testing.assert_array_equal(b.get('nefc')[:, 0], [o.nefc for o in ora])
  nsteps = 300
  if precision == 64:
    for _ in range(nsteps // 50):
      b.step(50)
      for o in ora:
        o.step(50)
      err = _rel_err(b.get('qpos'), np.array([o.qpos for o in ora]))
      assert err <= TOL_F64_1000, err
  else:
    # fp32: teacher-forced single steps from oracle states sampled along the trajectory
    worst = 0.0
    for k in range(20):
      for o in ora:
        o.step(10)
      qo, vo = np.array([o.qpos for o in ora]), np.array([o.qvel for o in ora])
      wo = np.array([o.qacc_warmstart for o in ora])
      b.set('qpos', qo); b.set('qvel', vo); b.set('qacc_warmstart', wo)
      b.step(1)
      for o in ora:
        o.step(1)
      worst = max(worst, _rel_err(b.get('qpos'), np.array([o.qpos for o in ora])))
    assert worst <= TOL_F32_ONE_STEP, worst
  assert not b.get('warning').any()
  b.close()


@pytest.mark.parametrize('precision,cone,condim', [(64, 'pyramidal', 3), (64, 'elliptic', 3), (64, 'elliptic', 6),
                                                   (64, 'pyramidal', 4), (32, 'elliptic', 4), (32, 'pyramidal', 3)])
def test_noslip_matches_oracle(precision, cone, condim):
  """noslip post-solver (composer/arena.xml:4: 5 sweeps) on the scene above: fp64 tracks the oracle over
  300 steps, fp32 one step at a time from the oracle's states."""
  xml = _ELLIPTIC_SCENE.format(condim=condim, impratio=1.0).replace('cone="elliptic"', 'cone="%s" noslip_iterations="5"' % cone)
  m = mc.compile_xml(xml)
  B = 16
  rs = np.random.RandomState(12)
  q = np.tile(m.qpos0, (B, 1))
  v = rs.uniform(-1, 1, (B, m.nv))
  b = _batch(m, B, precision=precision)
  b.set('qpos', q); b.set('qvel', v)
  ora = _oracles(m, q, v)
  if precision == 64:
    for _ in range(6):
      b.step(50)
      for o in ora:
        o.step(50)
      assert _rel_err(b.get('qpos'), np.array([o.qpos for o in ora])) <= TOL_F64_1000
    np.testing.assert_allclose(b.get('qacc'), np.array([o.qacc for o in ora]), rtol=0, atol=1e-6)
  else:
    worst = 0.0
    for k in range(20):
      for o in ora:
        o.step(10)
      b.set('qpos', np.array([o.qpos for o in ora])); b.set('qvel', np.array([o.qvel for o in ora]))
      b.set('qacc_warmstart', np.array([o.qacc_warmstart for o in ora]))
      b.step(1)
      for o in ora:
        o.step(1)
      worst = max(worst, _rel_err(b.get('qpos'), np.array([o.qpos for o in ora])))
    assert worst <= TOL_F32_ONE_STEP, worst
  assert not b.get('warning').any()
  b.close()


@pytest.mark.parametrize('precision,tol', [(64, 1e-10), (32, 5e-4)])
def test_plane_cylinder_contacts_match_oracle(precision, tol):
  # mjc_PlaneCylinder restated (up to 4 contacts); short horizon: a wobbling disc is chaotic
  m = mc.compile_xml("""<mujoco><option timestep="0.002"/><worldbody><geom type="plane" size="2 2 .1"/>
    <body pos="0 0 .2" quat="0.9 0.3 0.2 0.1"><freejoint/><geom type="cylinder" size=".1 .05" density="800"/></body>
    <body pos=".5 0 .06"><freejoint/><geom type="cylinder" size=".1 .05" density="800"/></body>
    <body pos="1 0 .11" quat="0.70710678118 0.70710678118 0 0"><freejoint/><geom type="cylinder" size=".1 .05" density="800"/></body>
  </worldbody></mujoco>""")
  B = 4
  q = np.tile(m.qpos0, (B, 1))
  b = _batch(m, B, precision=precision)
  b.set('qpos', q)
  ora = _oracles(m, q)
  b.step(250)
  for o in ora:
    o.step(250)
  np.testing.assert_array_equal(b.get('ncon')[:, 0], [o.ncon for o in ora])
  assert ora[0].ncon >= 5                      # 3 under the flat disc, 2 under the one on its side
  np.testing.assert_allclose(b.get('qpos'), np.array([o.qpos for o in ora]), rtol=0, atol=tol)
  assert not b.get('warning').any()
  b.close()


@pytest.mark.parametrize('precision,tol,lanes', [(64, 1e-9, 64), (64, 1e-9, 32), (32, 1e-3, 16)])
def test_box_piles_match_oracle(precision, tol, lanes):
  """sphere-box, capsule-box and box-box contacts (tests/test_box_collision.py): random piles, one per
  environment, while they form."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_box_collision import _scene
  rs = np.random.RandomState(5)
  bodies = []
  for k in range(5):
    gtype = ['box', 'box', 'capsule', 'sphere', 'box'][k]
    size = {'box': rs.uniform(.04, .12, 3), 'capsule': (rs.uniform(.03, .05), rs.uniform(.05, .1)), 'sphere': (rs.uniform(.04, .07),)}[gtype]
    bodies.append((gtype, size, (0, 0, .15 + .22*k), (1, 0, 0, 0)))
  m = mc.compile_xml(_scene(bodies, 'cone="elliptic"'))
  B = 8
  q = np.tile(m.qpos0, (B, 1))
  for e in range(B):
    for k in range(5):
      quat = rs.randn(4)
      q[e, 7*k + 3:7*k + 7] = quat / np.linalg.norm(quat)
      q[e, 7*k:7*k + 2] = rs.uniform(-.08, .08, 2)
  b = _batch(m, B, precision=precision, lanes_per_env=lanes, nconmax=32)
  b.set('qpos', q)
  b.step(300)
  # Five free bodies are up to five constraint islands: the fp64 batch solves per island like the oracle (StepCore::
  # solve_islands, on by default in fp64), the fp32 batch jointly (the same minimiser; 6e-9 apart over these 300 steps)
  ora = _oracles(m, q)
  for o in ora:
    o.step(300)
  np.testing.assert_array_equal(b.get('ncon')[:, 0] > 0, [o.ncon > 0 for o in ora])
  np.testing.assert_allclose(b.get('qpos'), np.array([o.qpos for o in ora]), rtol=0, atol=tol)
  assert not b.get('warning').any()
  b.close()


def test_elliptic_contact_force_equals_weight_on_gpu():
  # wrapper/core_test.py:393-416 with cone="elliptic", through touch (sums normal forces) and
  # qfrc_constraint; fp64 kernel.
  m = mc.compile_xml("""
  <mujoco><option cone="elliptic"/><worldbody>
    <geom name='floor' type='plane' size='1 1 1'/>
    <body name='box' pos='0 0 .1'><freejoint/>
      <geom name='box' type='box' size='.1 .1 .1'/>
      <site name='s' type='box' size='.11 .11 .11'/></body>
  </worldbody><sensor><touch name='t' site='s'/></sensor></mujoco>""")
  b = _batch(m, 4, precision=64)
  b.legacy_step = False
  b.step(500)
  b.forward()
  np.testing.assert_allclose(b.get('sensordata')[:, 0], 9.81 * m.body_mass[1], rtol=0, atol=1e-6)
  np.testing.assert_array_equal(b.get('nefc')[:, 0], 12)
  b.close()


@pytest.mark.parametrize('name,nsub', [('hopper', 4), ('humanoid', 5)])
def test_rollout_with_acceleration_stage_sensors(name, nsub):
  """Sensors are only evaluated in the passes whose values can be observed; with several
  substeps per env-step and acceleration-stage sensors (touch, accelerometer, force / torque) the
  one-launch rollout must still report exactly what the launch-per-step loop reports."""
  import torch
  m = _model(name)
  B, T = 8, 12
  rs = np.random.RandomState(4)
  q = np.tile(m.qpos0, (B, 1))
  if name == 'hopper':
    q[:, 1] -= 0.35                      # start touching the ground: touch sensors fire
  acts = rs.uniform(-1, 1, (T, B, m.nu))
  loop = _batch(m, B, precision=64)
  loop.set('qpos', q)
  want_q, want_s = [], []
  for t in range(T):
    loop.set_control(acts[t])
    loop.step(nsub)
    want_q.append(loop.get('qpos')); want_s.append(loop.get('sensordata'))
  assert np.abs(np.stack(want_s)).max() > 0
  ro = _batch(m, B, precision=64)
  ro.set('qpos', q)
  ctrl = torch.from_numpy(np.ascontiguousarray(acts.transpose(0, 2, 1))).to('cuda').contiguous()
  qs = torch.zeros((T, m.nq, B), dtype=torch.float64, device='cuda')
  ss = torch.zeros((T, m.nsensordata, B), dtype=torch.float64, device='cuda')
  ro.rollout(T, nsub, ctrl.data_ptr(), qs.data_ptr(), None, ss.data_ptr())
  ro.sync()
  np.testing.assert_array_equal(qs.cpu().numpy().transpose(0, 2, 1), np.stack(want_q))
  np.testing.assert_array_equal(ss.cpu().numpy().transpose(0, 2, 1), np.stack(want_s))
  # and against the oracle, which evaluates every sensor in every step
  ora = _oracles(m, q)
  from oracle import oracle
  for t in range(T):
    oracle.rollout_legacy(ora, acts[t][None], nsub=nsub)
  so = np.stack([o.sensordata for o in ora])
  np.testing.assert_allclose(want_s[-1], so, rtol=1e-7, atol=1e-7 * max(1.0, np.abs(so).max()))
  loop.close(); ro.close()


@pytest.mark.parametrize('seed,ellipsoids,noslip', [(s, False, 0) for s in range(24)] + [(s, True, 0) for s in range(8)] +
                         [(s, s % 2 == 1, 3) for s in range(8)])
def test_random_models_match_oracle(seed, ellipsoids, noslip):
  """Parity fuzzing (tests/random_models.py): random articulated models -- free / ball / hinge / slide
  joints, several roots, capsule / sphere / ellipsoid contacts, pyramidal and elliptic cones of every condim, Euler and
  RK4, fluid drag, motors / servos, random sensors -- fp64 kernel vs oracle, lane widths rotating with
  the seed.  Tolerance: the two solvers stop at MuJoCo's 1e-8 tolerance and may stop ~1e-8 apart in
  qacc where their iteration paths differ in the last bits (see tests/test_random_models.py)."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from random_models import random_model_xml
  m = mc.compile_xml(random_model_xml(seed, ellipsoids, noslip))
  B = 4
  rs = np.random.RandomState(1000 + seed)
  q = np.tile(m.qpos0, (B, 1))
  v = rs.uniform(-.5, .5, (B, m.nv))
  b = _batch(m, B, precision=64, lanes_per_env=(64, 32, 16)[seed % 3])
  b.set('qpos', q); b.set('qvel', v)
  ora = _oracles(m, q, v)
  from oracle import oracle
  for t in range(15):
    a = rs.uniform(-1, 1, (B, m.nu))
    b.set_control(a)
    b.step(10)
    oracle.rollout_legacy(ora, a[None], nsub=10)
  qo = np.stack([o.qpos for o in ora])
  assert _rel_err(b.get('qpos'), qo) <= 1e-6
  so = np.stack([o.sensordata for o in ora])
  np.testing.assert_allclose(b.get('sensordata'), so, rtol=0, atol=1e-5 * max(1.0, np.abs(so).max()))
  np.testing.assert_array_equal(b.get('warning').sum(axis=0), np.sum([o.warning for o in ora], axis=0))
  b.close()


@pytest.mark.parametrize('seed,noslip', [(s, 0) for s in range(200, 210)] + [(s, 3) for s in range(217, 221)])
def test_random_models_with_cylinders_match_oracle(seed, noslip):
  """The fuzzing above with three static cylinders under the trees: sphere-cylinder / capsule-cylinder contacts on the
  device (fp64 kernel vs oracle)."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from random_models import random_model_xml
  m = mc.compile_xml(random_model_xml(seed, False, noslip, cylinders=True))
  B = 4
  rs = np.random.RandomState(3000 + seed)
  q = np.tile(m.qpos0, (B, 1))
  v = rs.uniform(-.5, .5, (B, m.nv))
  b = _batch(m, B, precision=64, lanes_per_env=(64, 32, 16)[seed % 3])
  b.set('qpos', q); b.set('qvel', v)
  ora = _oracles(m, q, v)
  from oracle import oracle
  for t in range(15):
    a = rs.uniform(-1, 1, (B, m.nu))
    b.set_control(a)
    b.step(10)
    oracle.rollout_legacy(ora, a[None], nsub=10)
  qo = np.stack([o.qpos for o in ora])
  assert _rel_err(b.get('qpos'), qo) <= 1e-6
  np.testing.assert_array_equal(b.get('warning').sum(axis=0), np.sum([o.warning for o in ora], axis=0))
  b.close()


# ---- the HBM stash of the position / velocity stage between legacy steps --------------------------
@pytest.mark.parametrize('name,precision,lanes', [('cheetah', 32, 32), ('cheetah', 64, 16), ('cartpole', 32, 32),
                                                  ('hopper', 32, 32), ('humanoid', 32, 64)])
def test_stash_between_legacy_steps_is_bit_identical_on_gpu(name, precision, lanes):
  """What mjData keeps between the mj_step1 that ends one legacy Physics.step() and the mj_step2 that begins
  the next (engine.py:147-162) is kept in a per-env HBM stash; a launch that reloads it must produce the very
  same bits as one that recomputes the stage, and every state edit must invalidate it."""
  import torch
  m = _model(name)
  B, T = 37, 60
  rs = np.random.RandomState(3)
  q = np.tile(m.qpos0, (B, 1))
  k = min(3, m.nq)
  q[:, -k:] += rs.uniform(-.2, .2, (B, k))
  a, b = _batch(m, B, precision=precision, lanes_per_env=lanes), _batch(m, B, precision=precision, lanes_per_env=lanes)
  a.set_opt('stash', 0)
  b.set_opt('stash', 1)
  assert a.info()['stash'] == 0 and b.info()['stash'] == 1
  td = torch.float32 if precision == 32 else torch.float64
  qdevs = [torch.zeros((m.nq, B), dtype=td, device='cuda') for _ in range(2)]
  for e in (a, b):
    e.set('qpos', q)
  for t in range(T):
    c = rs.uniform(-1, 1, (B, m.nu))
    for e in (a, b):
      e.set_control(c)
      e.step(1 + t % 3)
    for f in ('qpos', 'qvel', 'sensordata', 'xpos', 'qacc_warmstart', 'time'):
      np.testing.assert_array_equal(a.get(f), b.get(f), err_msg='%s at step %d' % (f, t))
    if t == 20:        # edit through the C-ABI: invalidates by itself
      for e in (a, b):
        e.set('qvel', np.zeros((B, m.nv)))
    if t == 30:        # masked reset of some environments
      mask = (np.arange(B) % 3 == 0).astype(np.int32)
      for e in (a, b):
        e.reset(mask)
    if t == 40:        # edit through bound device memory: the caller must invalidate
      for e, qdev in zip((a, b), qdevs):
        qdev.copy_(torch.from_numpy(np.ascontiguousarray(q.T)).to(td))
        e.sync()
        e.bind('qpos', qdev.data_ptr())
        e.invalidate()
        e.step()
        e.sync()
        got = e.get('qpos')
        e.bind('qpos', 0)
        e.set('qpos', got)
  assert np.abs(a.get('qpos') - q).max() > 1e-3
  a.close(); b.close()


# ---- equality constraints between bodies and joints on the HIP path --------------------------------
@pytest.mark.parametrize('precision,tol,lanes', [(64, 1e-10, 64), (64, 1e-10, 16), (32, 5e-4, 32)])
def test_connect_weld_joint_equalities_match_oracle_on_gpu(precision, tol, lanes):
  """connect (3 rows), weld (6 rows incl. the quaternion-error rows) and joint-coupling equalities through the
  C-ABI against the oracle (the host-build twin of this test: tests/test_kernel_logic_emu.py)."""
  import test_oracle_kat as kat
  from oracle.oracle import OraclePhysics
  scenes = [kat._EQ_CHAIN,
            """<mujoco><option timestep="0.002"/><worldbody><geom type="plane" size="2 2 .1"/>
               <body name="box" pos=".3 .1 .7" quat=".8 .2 .4 .1"><freejoint/><geom type="box" size=".1 .05 .02" mass="2"/></body>
               <body name="b2" pos=".6 .1 .7"><freejoint/><geom type="sphere" size=".05"/></body>
               <body name="bob" pos="0 0 1"><freejoint/><geom type="sphere" size=".02" mass="1"/></body></worldbody>
               <equality><weld body1="box" body2="b2" anchor="-.15 0 0"/><connect body1="bob" anchor="0 0 .5" solref="0.004 1"/></equality></mujoco>"""]
  for xml in scenes:
    m = mc.compile_xml(xml)
    B = 5
    rs = np.random.RandomState(1)
    v = rs.uniform(-.5, .5, (B, m.nv))
    b = _batch(m, B, precision=precision, lanes_per_env=lanes)
    b.set('qvel', v)
    refs = []
    for e in range(B):
      o = OraclePhysics(m)
      o.qvel[:] = v[e]
      o.forward()
      refs.append(o)
    b.forward()
    assert (b.get('nefc')[:, 0] == [o.nefc for o in refs]).all() and refs[0].nefc >= 9
    for _ in range(300):
      b.step()
      for o in refs:
        o.step()
    qo = np.stack([o.qpos for o in refs])
    np.testing.assert_allclose(b.get('qpos'), qo, rtol=0, atol=tol)
    assert not b.get('warning').any()
    b.close()


@pytest.mark.parametrize('name,precision,lanes', [('cheetah', 64, 32), ('cheetah', 32, 32), ('humanoid', 32, 64)])
def test_step1_step2_entry_points_on_gpu(name, precision, lanes):
  """dmc_batch_step1 / dmc_batch_step2 (mujoco.mj_step1 / mj_step2 on their own, engine.py:156-162): the pair is
  bit-identical to one non-legacy mj_step launch, the arrays step1 writes are those of the current state (vs the
  oracle's mj_step1), and a state edit between the two makes step2 recompute the stage."""
  from oracle.oracle import OraclePhysics, OracleModel
  m = _model(name)
  B, T = 9, 40
  rs = np.random.RandomState(8)
  q = np.tile(m.qpos0, (B, 1))
  q[:, -3:] += rs.uniform(-.2, .2, (B, 3))
  a, b = _batch(m, B, precision=precision, lanes_per_env=lanes), _batch(m, B, precision=precision, lanes_per_env=lanes)
  a.legacy_step = False
  om = OracleModel(m)
  refs = [OraclePhysics(om) for _ in range(B)]
  for e in (a, b):
    e.set('qpos', q)
  for k, o in enumerate(refs):
    o.qpos[:] = q[k]
  tol = 1e-10 if precision == 64 else 2e-4
  for t in range(T):
    c = rs.uniform(-1, 1, (B, m.nu))
    for e in (a, b):
      e.set_control(c)
    a.step()
    b.step1()
    xg = b.get('xpos')
    for k, o in enumerate(refs):
      o.ctrl[:] = c[k]
      if precision == 32 and t:      # teacher-forced: the fp32 trajectory drifts from the fp64 one
        o.qpos[:] = qprev[k]; o.qvel[:] = vprev[k]; o.qacc_warmstart[:] = wprev[k]
      o.step1()
      np.testing.assert_allclose(xg[k], np.array(o.xpos), rtol=0, atol=tol)
    b.step2()
    for o in refs:
      o.step2()
    qprev, vprev, wprev = b.get('qpos'), b.get('qvel'), b.get('qacc_warmstart')
    np.testing.assert_array_equal(a.get('qpos'), qprev, err_msg='step %d' % t)
    np.testing.assert_array_equal(a.get('qvel'), vprev)
    np.testing.assert_allclose(qprev, np.stack([o.qpos for o in refs]), rtol=0, atol=tol)
    if t == 20:
      b.step1()
      for e in (a, b):
        e.set('qvel', np.zeros((B, m.nv)))
      a.step()
      b.step2()
      np.testing.assert_array_equal(a.get('qpos'), b.get('qpos'))
      qprev, vprev, wprev = b.get('qpos'), b.get('qvel'), b.get('qacc_warmstart')
      for k, o in enumerate(refs):
        o.qpos[:] = qprev[k]; o.qvel[:] = vprev[k]; o.qacc_warmstart[:] = wprev[k]
  a.close(); b.close()


@pytest.mark.parametrize('precision,tol', [(64, 1e-10), (32, 2e-4)])
def test_xfrc_applied_on_gpu(precision, tol):
  """mjData.xfrc_applied through the C-ABI field: per-env Cartesian wrenches on three bodies of the humanoid against
  the oracle (the host-build twin: tests/test_kernel_logic_emu.py::test_xfrc_applied_matches_oracle)."""
  from oracle.oracle import OraclePhysics, OracleModel
  m = _model('humanoid')
  B = 4
  rs = np.random.RandomState(3)
  x = np.zeros((B, m.nbody, 6))
  for e in range(B):
    for bd in rs.choice(np.arange(1, m.nbody), 3, replace=False):
      x[e, bd] = rs.uniform(-1, 1, 6) * [30, 30, 60, 3, 3, 3]
  x[3] = 0                                  # one env without any wrench
  b = _batch(m, B, precision=precision, nconmax=24)
  b.set('xfrc_applied', x.reshape(B, -1))
  om = OracleModel(m)
  refs = [OraclePhysics(om) for _ in range(B)]
  for e, o in enumerate(refs):
    o.xfrc_applied[:] = x[e].ravel()
    o.forward()
  for t in range(120):
    c = rs.uniform(-1, 1, (B, m.nu))
    if precision == 32 and t:
      b.set('qpos', np.stack([o.qpos for o in refs])); b.set('qvel', np.stack([o.qvel for o in refs]))
      b.set('qacc_warmstart', np.stack([o.qacc_warmstart for o in refs]))
    b.set_control(c)
    b.step()
    for e, o in enumerate(refs):
      o.ctrl[:] = c[e]
      o.step()
    qo = np.stack([o.qpos for o in refs])
    np.testing.assert_allclose(b.get('qpos'), qo, rtol=0, atol=tol * max(1.0, np.abs(qo).max()), err_msg='step %d' % t)
  if precision == 64:
    so = np.stack([o.sensordata for o in refs])
    np.testing.assert_allclose(b.get('sensordata'), so, rtol=0, atol=1e-7 * max(1.0, np.abs(so).max()))
  b.reset()
  assert not b.get('xfrc_applied').any()            # mj_resetData zeroes it
  b.close()


@pytest.mark.parametrize('asset,nsub', [('cheetah', 1), ('humanoid', 2)])
def test_kinematic_stash_bit_identical_and_notices_silent_edits(asset, nsub, monkeypatch):
  """The kinematic stash (on by default) keeps poses / COM frame / velocities between legacy steps together with the
  (qpos, qvel) they belong to.  Same trajectories bit for bit as without it, including after the state was rewritten
  directly in device memory (a torch tensor, no dmc_batch_set, no invalidate)."""
  import torch
  m = _model(asset)
  B = 64
  rs = np.random.RandomState(1)
  q = np.tile(m.qpos0, (B, 1))
  q[:, -4:] += rs.uniform(-0.2, 0.2, (B, 4))
  acts = rs.uniform(-1, 1, (12, B, m.nu))
  out = {}
  for mode in ('on', 'off'):
    monkeypatch.delenv('DMC_NO_KSTASH', raising=False)
    if mode == 'off':
      monkeypatch.setenv('DMC_NO_KSTASH', '1')
    b = _batch(m, B, precision=32)
    b.set('qpos', q)
    rows = []
    qdev = torch.empty((m.nq, B), dtype=torch.float32, device='cuda')      # SoA: (rows, B)
    for t, a in enumerate(acts):
      if t == 5:
        # from here on qpos lives in a torch tensor; its first write is a silent edit of the state
        qdev.copy_(torch.as_tensor(b.get('qpos').T.copy(), device='cuda'))
        b.bind('qpos', qdev.data_ptr())
      if t == 8:
        qdev[-1, :] += 0.05
        torch.cuda.synchronize()
      b.set_control(a)
      b.step(nsub)
      rows.append((b.get('qpos').copy(), b.get('qvel').copy(), b.get('sensordata').copy(), b.get('xpos').copy()))
    out[mode] = rows
    b.close()
  for t, (x, y) in enumerate(zip(out['on'], out['off'])):
    for u, v in zip(x, y):
      np.testing.assert_array_equal(u, v, err_msg='step %d' % t)


@pytest.mark.parametrize('asset,precision,teacher,tol', [('cheetah', 64, False, 1e-9), ('humanoid', 64, False, 1e-5),
                                                         ('cheetah', 32, True, 1e-3), ('humanoid', 32, True, 2e-3)])
def test_cg_solver_on_device(asset, precision, teacher, tol):
  """option solver="CG" on the device (generic kernel: the baked layouts are Newton's) against the oracle.  CG stops at
  the solver tolerance instead of converging quadratically, so two implementations agree per step to ~1e-8 of the
  acceleration scale, not to rounding: fp64 open loop over 60 steps (measured 8e-15 cheetah, 1.6e-6 humanoid), fp32
  teacher-forced (measured 1.1e-4 / 5.6e-4 per step: in fp32 the conjugate directions lose orthogonality to rounding
  and the iteration stops on its floors; the production solver is Newton)."""
  from oracle import oracle
  with open(os.path.join(ASSETS, asset + '.xml')) as f:
    xml = f.read()
  m = mc.compile_xml(xml.replace('<option', '<option solver="CG" iterations="100"', 1))
  B, T = 16, 60
  rs = np.random.RandomState(5)
  q = np.tile(m.qpos0, (B, 1))
  q[:, -3:] += rs.uniform(-0.3, 0.3, (B, 3))
  b = _batch(m, B, precision=precision)
  assert b.info()['static_id'] == -1
  b.set('qpos', q)
  refs = _oracles(m, q)
  worst, iters = 0.0, 0
  for t in range(T):
    a = rs.uniform(-1, 1, (B, m.nu))
    if teacher:
      b.set('qpos', np.stack([p.qpos for p in refs]))
      b.set('qvel', np.stack([p.qvel for p in refs]))
      b.set('qacc_warmstart', np.stack([p.qacc_warmstart for p in refs]))
    b.set_control(a)
    b.step()
    oracle.rollout_legacy(refs, a[None])
    worst = max(worst, _rel_err(b.get('qpos'), np.stack([p.qpos for p in refs])))
    iters = max(iters, int(b.get('solver_iter').max()))
  print('measured: cg %s fp%d %s max rel dqpos=%.3g, max iterations %d' % (asset, precision, 'teacher-forced' if teacher else 'open loop', worst, iters))
  assert worst < tol, worst
  assert iters > 4 and not b.get('warning').any()
  b.close()


@pytest.mark.parametrize('cone,condim,precision,tol', [('pyramidal', 3, 64, 1e-9), ('elliptic', 3, 64, 1e-9), ('elliptic', 4, 64, 1e-9),
                                                       ('elliptic', 6, 64, 1e-9), ('pyramidal', 3, 32, 1e-4), ('elliptic', 4, 32, 1e-4)])
def test_pgs_solver_on_device(cone, condim, precision, tol):
  """option solver="PGS" (north_star "PGS/Newton"; the reference's own hot-path test asset
  mujoco/testing/assets/humanoid.xml:9 = suite/assets/testing_humanoid_pgs.xml: PGS + RK4) on the device against the
  oracle, teacher-forced including the warm start (PGS stops at its 50-iteration cap, far from convergence: its answer
  depends on the warm start and open-loop runs separate by chaos).  Pyramidal and elliptic cones, condim 3 / 4 / 6."""
  from oracle import oracle
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_pgs import _xml
  m = mc.compile_xml(_xml(cone=cone, floor_condim=condim))
  assert m.opt.solver == 0 and m.opt.integrator == 1
  B, T = 16, 120
  rs = np.random.RandomState(9)
  q = np.tile(m.qpos0, (B, 1))
  q[:, 2] = rs.uniform(0.5, 1.0, B)
  q[:, 7:] += rs.uniform(-0.3, 0.3, (B, m.nq - 7))
  b = _batch(m, B, precision=precision)
  refs = _oracles(m, q)
  worst, iters, nefc = 0.0, 0, 0
  for t in range(T):
    a = rs.uniform(-1, 1, (B, m.nu)).astype(np.float32).astype(np.float64)
    b.set('qpos', np.stack([p.qpos for p in refs]))
    b.set('qvel', np.stack([p.qvel for p in refs]))
    b.set('qacc_warmstart', np.stack([p.qacc_warmstart for p in refs]))
    b.set_control(a)
    b.step()
    oracle.rollout_legacy(refs, a[None])
    worst = max(worst, _rel_err(b.get('qpos'), np.stack([p.qpos for p in refs])))
    iters = max(iters, int(b.get('solver_iter').max()))
    nefc = max(nefc, int(b.get('nefc').max()))
    if precision == 64:
      np.testing.assert_array_equal(b.get('solver_iter')[:, 0], [p.solver_iter for p in refs])
  print('measured: pgs %s condim %d fp%d teacher-forced max rel dqpos=%.3g, max iterations %d, max nefc %d' % (cone, condim, precision, worst, iters, nefc))
  assert worst < tol, worst
  assert iters >= 10 and nefc >= 12 and not b.get('warning').any()
  b.close()


@pytest.mark.parametrize('precision', [32, 64])
def test_async_host_transfers_equal_the_synchronous_ones(cheetah, precision):
  """dmc_batch_set_async / get_async / get_wait (pinned staging, device-side transposition, one copy for several fields,
  fp32 on the wire) against dmc_batch_set / dmc_batch_get: same device contents, same host values; more sets in a row
  than staging slots; trajectories driven through either boundary are identical."""
  m = cheetah
  B = 37                                   # not a multiple of the transposition tile
  a, b = _batch(m, B, precision=precision), _batch(m, B, precision=precision)
  rs = np.random.RandomState(0)
  q = _cheetah_init(m, B)
  a.set('qpos', q); b.set_async('qpos', q)
  for hdt in (np.float64, np.float32):
    for t in range(9):                     # > 4 staging slots
      c = rs.uniform(-1, 1, (B, m.nu)).astype(hdt)
      a.set('ctrl', c); b.set_async('ctrl', c)
      a.step(); b.step()
      got = b.get_many(('qpos', 'qvel', 'sensordata', 'time'), dtype=hdt)
      for n in ('qpos', 'qvel', 'sensordata', 'time'):
        want = a.get(n)
        assert got[n].dtype == hdt and got[n].shape == want.shape
        np.testing.assert_array_equal(got[n], want.astype(hdt), err_msg=n)
  z = b.get_many(('qpos', 'time'), copy=False)      # views of the pinned staging, in the batch's own precision
  assert z['qpos'].dtype == (np.float64 if precision == 64 else np.float32) and z['time'].dtype == np.float64 and not z['qpos'].flags.writeable
  np.testing.assert_array_equal(z['qpos'], a.get('qpos').astype(z['qpos'].dtype))
  np.testing.assert_array_equal(z['time'], a.get('time'))
  # an enqueued get is a snapshot of its point in the stream: a later step does not change what it returns
  before = a.get('qpos')
  b.get_async(('qpos',))
  b.step()
  np.testing.assert_array_equal(b.get_wait()['qpos'], before)
  with pytest.raises(Exception, match='no get is enqueued'):
    b._pending_get = ['qpos']; b.get_wait()
  a.close(); b.close()


def test_optional_launch_features_on_a_small_specialised_model(cheetah):
  """The fp32 kernels specialised for the small suite models are built without the optional launch features (substep
  probe, legacy_step 2 = step + mj_forward, implicitfast: step_core.h kFeat): a launch that needs one runs the generic
  kernel on the same batch -- same state layout, results equal to rounding (the two kernels order some sums differently)."""
  import torch
  m = cheetah
  B = 16
  q = _cheetah_init(m, B)
  a, b = _batch(m, B, precision=32), _batch(m, B, precision=32)
  assert a.info()['static_id'] >= 0
  probe = torch.zeros((4, 3, B), dtype=torch.float32, device='cuda')
  g = m.name2id('ffoot', 'geom')
  b.set_step_probe(g, probe.data_ptr(), 4)
  rs = np.random.RandomState(0)
  for x in (a, b):
    x.set('qpos', q)
  for t in range(5):
    c = rs.uniform(-1, 1, (B, m.nu))
    a.set('ctrl', c); b.set('ctrl', c)
    trace = []
    for k in range(3):
      a.step(1)
      trace.append(a.get('geom_xpos')[:, 3*g:3*g + 3])
    a.forward()
    b.step(3, forward_after=True)
    torch.cuda.synchronize()
    for n in ('qpos', 'qvel', 'qacc', 'sensordata'):
      np.testing.assert_allclose(b.get(n), a.get(n), rtol=0, atol=2e-4 * max(1.0, np.abs(a.get(n)).max()), err_msg=n)
    np.testing.assert_allclose(probe[:3].cpu().numpy().transpose(0, 2, 1), np.stack(trace), rtol=0, atol=1e-5)
  a.close(); b.close()


_STATIC_WORLD = """<mujoco><worldbody><geom name='f' type='plane' size='1 1 .1'/>
<body name='post' pos='0 0 1' euler='0 30 0'><geom type='sphere' size='.1'/><site name='s' pos='.1 0 0'/></body></worldbody>
<sensor><framepos objtype='site' objname='s'/><framezaxis objtype='site' objname='s'/></sensor></mujoco>"""


@pytest.mark.parametrize('xml', ['<mujoco/>', _STATIC_WORLD], ids=['empty', 'static_world'])
@pytest.mark.parametrize('precision', [64, 32])
def test_models_without_degrees_of_freedom_step_on_the_device(xml, precision):
  """MuJoCo steps a model with nv = 0 (the reference's composer tests build such arenas: composer/environment_test.py):
  time advances, the poses and position sensors are evaluated, nothing else moves."""
  from oracle.oracle import OraclePhysics
  m = mc.compile_xml(xml)
  assert m.nv == 0
  b = _batch(m, 3, precision=precision)
  b.forward(False)
  b.step(4)
  o = OraclePhysics(m)
  o.forward()
  o.step(4)
  np.testing.assert_allclose(b.get('time').ravel(), [o.time] * 3, rtol=0, atol=1e-12)
  assert not b.get('warning').any() and not b.get('ncon').any()
  tol = 1e-12 if precision == 64 else 1e-6
  np.testing.assert_allclose(b.get('xpos')[0], np.asarray(o.xpos).ravel(), rtol=0, atol=tol)
  if m.nsensordata:
    np.testing.assert_allclose(b.get('sensordata')[0], np.asarray(o.sensordata).ravel(), rtol=0, atol=tol)


def test_config4_model_holds_five_environments_per_cu():
  """Offload level 3 + five-wave workgroups: the 62-dof walker of BASELINE config 4 at its production caps
  keeps FIVE environments resident per CU in fp32 (a field added to the LDS scratch would silently cost the fifth: -8 %)."""
  m = _model('cmu_2019_position_floor')
  b = _batch(m, 4096, precision=32, nconmax=48)      # (a batch that fills the chip: small ones are spread one wave per CU)
  info = b.info()
  assert info['static_id'] >= 0 and info['envs_per_cu'] == 5 and info['waves_per_block'] == 5, info
  assert info['lds_bytes_per_block'] <= 160 * 1024
  b.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name, nconmax', [('soccer_2v2_boxhead', 24), ('humanoid', 24)])
def test_a_small_batch_keeps_its_contact_rows_in_lds(name, nconmax, monkeypatch):
  """17 .. 32 dofs: the contact rows and the kept factor of M live in the per-env global scratch so that more environments
  are resident per CU -- unless the batch is at most one environment per CU AND a model-specialised kernel exists for
  the all-in-LDS layout (dmc_api.hip; caps[4] / DMC_JLEVEL force either).  Only WHERE the rows are stored differs: the
  trajectories are bit-identical."""
  m = _model(name)
  B, T = 64, 40
  rs = np.random.RandomState(3)
  acts = rs.uniform(-1, 1, (T, B, m.nu))
  monkeypatch.delenv('DMC_JLEVEL', raising=False)
  b = _batch(m, B, precision=32, nconmax=nconmax)
  info = b.info()
  b.close()
  assert info['static_id'] >= 0, info      # (the small batch never trades a baked kernel for the generic one)
  if name == 'soccer_2v2_boxhead': assert info['global_scratch_bytes_per_env'] == 0, info
  big = _batch(m, 4096, precision=32, nconmax=nconmax)
  assert big.info()['global_scratch_bytes_per_env'] > 0
  big.close()
  out = {}
  if name == 'humanoid': monkeypatch.setenv('DMC_NO_STATIC', '1')      # (only level 1 is baked for it: compare the generic kernel with itself)
  for level in ('0', '1'):
    monkeypatch.setenv('DMC_JLEVEL', level)
    b = _batch(m, B, precision=32, nconmax=nconmax)
    assert (b.info()['global_scratch_bytes_per_env'] == 0) == (level == '0'), b.info()
    b.forward()
    for t in range(T):
      b.set_control(acts[t]); b.step(2)
    out[level] = (b.get('qpos'), b.get('qvel'), b.get('sensordata'), b.get('warning'))
    b.close()
  for x, y in zip(out['0'], out['1']):
    assert np.array_equal(x, y)
