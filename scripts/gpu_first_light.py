"""First GPU run: stage-by-stage parity of the HIP kernel vs the fp64 oracle,
short trajectories, and a first timing.  Writes gpurun_out/first_light.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc  # noqa
from dm_control_amd.batch import BatchedPhysics  # noqa
from oracle.oracle import OraclePhysics  # noqa

out = {}
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/cheetah.xml')).read())
NE = 16


def init_states(rs, n):
  q = np.tile(m.qpos0, (n, 1))
  q[:, 3:] += rs.uniform(-0.4, 0.4, (n, 6))
  q[:, 2] += rs.uniform(-0.5, 0.5, n)
  q[:, 1] = rs.uniform(-0.6, 0.1, n)   # some envs start in penetration
  v = rs.uniform(-1, 1, (n, m.nv))
  return q, v


def stage_check(prec, lpe):
  rs = np.random.RandomState(0)
  q, v = init_states(rs, NE)
  c = rs.uniform(-1, 1, (NE, m.nu))
  b = BatchedPhysics(m, NE, precision=prec, lanes_per_env=lpe)
  b.debug_enable(NE)
  b.set('qpos', q); b.set('qvel', v); b.set('ctrl', c)
  b.forward()
  res = {}
  worst = {}
  for e in range(NE):
    o = OraclePhysics(m)
    o.qpos[:] = q[e]; o.qvel[:] = v[e]; o.ctrl[:] = c[e]
    o.forward()
    ne = o.nefc
    pairs = [('xpos', o.xpos), ('xmat', o.xmat), ('subtree_com', o.subtree_com), ('cinert', o.cinert),
             ('cdof', o.cdof), ('qM', o.qM), ('qfrc_bias', o.qfrc_bias), ('qfrc_passive', o.qfrc_passive),
             ('qacc_smooth', o.qacc_smooth), ('efc_J', o.efc_J[:ne*m.nv]), ('efc_D', o.efc_D[:ne]),
             ('efc_aref', o.efc_aref[:ne]), ('efc_force', o.efc_force[:ne]), ('qacc', o.qacc),
             ('qfrc_constraint', o.qfrc_constraint)]
    for name, ref in pairs:
      got = b.debug_get(name, e)[:ref.size]
      err = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max())) if ref.size else 0.0
      worst[name] = max(worst.get(name, 0.0), err)
    ncon = int(b.debug_get('imisc', e)[0]); nefc = int(b.debug_get('imisc', e)[1])
    if ncon != o.ncon or nefc != o.nefc:
      res.setdefault('count_mismatch', []).append((e, ncon, o.ncon, nefc, o.nefc))
  res['worst_rel_err'] = worst
  res['info'] = b.info()
  b.close()
  return res


def traj_check(prec, lpe, T=300):
  rs = np.random.RandomState(1)
  q, v = init_states(rs, NE)
  q[:, 1] = rs.uniform(-0.1, 0.1, NE)
  b = BatchedPhysics(m, NE, precision=prec, lanes_per_env=lpe)
  b.set('qpos', q); b.set('qvel', v * 0)
  os_ = [OraclePhysics(m) for _ in range(NE)]
  for e, o in enumerate(os_):
    o.qpos[:] = q[e]; o.forward()
  errs = []
  sens = []
  for t in range(T):
    c = rs.uniform(-1, 1, (NE, m.nu))
    b.set_control(c)
    b.step()
    for e, o in enumerate(os_):
      o.ctrl[:] = c[e]; o.step()
    qg = b.get('qpos')
    qo = np.stack([o.qpos for o in os_])
    errs.append(float((np.abs(qg - qo).max(axis=1) / np.maximum(1, np.abs(qo).max(axis=1))).max()))
    if t == T - 1:
      sg = b.get('sensordata'); so = np.stack([o.sensordata for o in os_])
      sens = float(np.abs(sg - so).max())
  w = b.get('warning').sum(axis=0).tolist()
  b.close()
  return dict(err_t1=errs[0], err_t10=errs[9], err_t100=errs[99], err_final=errs[-1], err_max=max(errs),
              sensor_err_final=sens, warnings=w, maxcon=int(max(o.ncon for o in os_)))


def timing(prec, lpe, B=4096, nstep=1, reps=30):
  rs = np.random.RandomState(2)
  b = BatchedPhysics(m, B, precision=prec, lanes_per_env=lpe)
  q = np.tile(m.qpos0, (B, 1)); q[:, 3:] += rs.uniform(-0.3, 0.3, (B, 6))
  b.set('qpos', q)
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  b.set_output_mask(1)
  b.step(50); b.sync()          # settle into contact
  ms = b.time_steps(nstep, reps)
  info = b.info()
  it = b.get('solver_iter').mean(); nc = b.get('ncon').mean()
  b.close()
  return dict(ms_per_launch=ms, steps_per_s=B * nstep / (ms * 1e-3), info=info, mean_iter=float(it), mean_ncon=float(nc))


t0 = time.time()
for prec in (64, 32):
  for lpe in (64, 32, 16):
    key = 'f%d_lpe%d' % (prec, lpe)
    try:
      out['stage_' + key] = stage_check(prec, lpe)
      out['traj_' + key] = traj_check(prec, lpe)
    except Exception as ex:  # pylint: disable=broad-except
      out['error_' + key] = repr(ex)
    print(key, json.dumps(out.get('stage_' + key, {}).get('worst_rel_err', {})), flush=True)
    print(key, json.dumps(out.get('traj_' + key, out.get('error_' + key))), flush=True)
for prec in (32, 64):
  for lpe in (64, 32, 16):
    for nstep in (1, 10):
      key = 'time_f%d_lpe%d_n%d' % (prec, lpe, nstep)
      try:
        out[key] = timing(prec, lpe, nstep=nstep)
      except Exception as ex:  # pylint: disable=broad-except
        out[key] = repr(ex)
      print(key, json.dumps(out[key]), flush=True)
out['wall_s'] = time.time() - t0
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'first_light.json'), 'w'), indent=1)
