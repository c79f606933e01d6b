"""GPU perf probe: times the cheetah B=4096 step for several (lanes, caps) and
reports contact/row statistics of the random-action workload."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc  # noqa
from dm_control_amd.batch import BatchedPhysics, OUT  # noqa

m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/cheetah.xml')).read())
B = int(os.environ.get('B', 4096))
lim = m.jnt_limited == 1
lo, hi = m.jnt_range[lim].T
q0 = np.tile(m.qpos0, (B, 1))
for e in range(B):
  q0[e, lim] = np.random.RandomState(e).uniform(lo, hi)
rs = np.random.RandomState(5)


def probe(prec, lanes, nconmax, njmax, nstep=1, reps=50, stats=False):
  b = BatchedPhysics(m, B, precision=prec, lanes_per_env=lanes, nconmax=nconmax, njmax=njmax)
  b.set('qpos', q0)
  b.set_output_mask(OUT['sensor'])
  b.step(200); b.sync()
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  res = dict(prec=prec, lanes=lanes, nstep=nstep, info=b.info())
  if stats:
    mc_, me_, mi_ = 0, 0, 0
    hist = np.zeros(64, int)
    for t in range(300):
      b.set_control(rs.uniform(-1, 1, (B, m.nu)))
      b.step()
      nc = b.get('ncon')[:, 0]; ne = b.get('nefc')[:, 0]; it = b.get('solver_iter')[:, 0]
      mc_, me_, mi_ = max(mc_, nc.max()), max(me_, ne.max()), max(mi_, it.max())
      hist += np.bincount(nc, minlength=64)[:64]
    res.update(max_ncon=int(mc_), max_nefc=int(me_), max_iter=int(mi_), ncon_hist=hist[:25].tolist(),
               warnings=b.get('warning').sum(axis=0).tolist())
  ms = b.time_steps(nstep, reps)
  res.update(ms_per_launch=ms, steps_per_s=B * nstep / (ms * 1e-3))
  if os.environ.get('ROLLOUT'):
    import torch, time
    T = 200
    td = torch.float32 if prec == 32 else torch.float64
    ctrl = (torch.rand((T, m.nu, B), device='cuda', dtype=td) * 2 - 1)
    ss = torch.zeros((T, m.nsensordata, B), dtype=td, device='cuda')
    qs = torch.zeros((T, m.nq, B), dtype=td, device='cuda'); vs = torch.zeros((T, m.nv, B), dtype=td, device='cuda')
    b.rollout(T, 1, ctrl.data_ptr(), qs.data_ptr(), vs.data_ptr(), ss.data_ptr()); b.sync()
    t0 = time.perf_counter(); b.rollout(T, 1, ctrl.data_ptr(), qs.data_ptr(), vs.data_ptr(), ss.data_ptr()); b.sync(); dt = time.perf_counter() - t0
    res.update(rollout_steps_per_s=B * T / dt)
  b.close()
  return res


out = []
cfgs = json.loads(os.environ.get('CFGS', '[]')) or [
    [32, 64, 0, 0, 1, True], [32, 64, 12, 54, 1, False], [32, 64, 8, 38, 1, False],
    [32, 32, 12, 54, 1, False], [32, 32, 8, 38, 1, False], [32, 16, 8, 38, 1, False],
    [32, 32, 8, 38, 10, False], [64, 64, 8, 38, 1, False], [64, 32, 8, 38, 1, False]]
for c in cfgs:
  try:
    r = probe(c[0], c[1], c[2], c[3], nstep=c[4], stats=c[5])
  except Exception as ex:  # pylint: disable=broad-except
    r = dict(cfg=c, error=repr(ex))
  print(json.dumps(r), flush=True)
  out.append(r)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'perf_probe_%s.json' % os.environ.get('TAG', 'x')), 'w'), indent=1)
