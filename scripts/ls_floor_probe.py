"""fp32 line-search slope floor (step_core.h primal_search, DMC_LS_SLOPE_ULPS): the emulated fp32 kernel core forced onto
the fp64 oracle's trajectory -- one physics step from the oracle's state, the SAME states for every setting -- error of the
step against the oracle's, cost evaluations per line search and Newton iterations per step.  CPU only.
  CONFIG=4 ENVS=4 STEPS=20 python scripts/ls_floor_probe.py 0 4 16 64"""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from emu_lib import EmuPhysics, lib
from oracle.oracle import OraclePhysics
from dm_control_amd.suite import common
cid = int(os.environ.get('CONFIG', 4))
cfg = bench.CONFIGS[cid]
m = bench.load_model(cfg['asset'])
caps = {k: v for k, v in common.DEFAULT_CAPS.get(cfg['asset'], {}).items() if k in ('nconmax', 'njmax', 'njcon')}
NE, T = int(os.environ.get('ENVS', 4)), int(os.environ.get('STEPS', 20))
if cid == 3:
  q0 = np.tile(m.qpos0, (NE, 1)); q0[:, 2] = 0.5; q0[:, 7:] += np.random.RandomState(3).uniform(-.3, .3, (NE, m.nq - 7))
else:
  q0 = bench.initial_qpos(cfg, m, NE, 0)
# the oracle's trajectory: state before every physics step and the state after it
traj = []
for env in range(NE):
  o = OraclePhysics(m)
  o.qpos[:] = q0[env]
  rs = np.random.RandomState(100 + env)
  for t in range(T + 3):
    o.ctrl[:] = rs.uniform(-1, 1, m.nu)
    for k in range(cfg['nsub']):
      before = (o.qpos.copy(), o.qvel.copy(), o.qacc_warmstart.copy(), o.ctrl.copy(), o.act.copy() if m.na else None)
      o.step()
      if t >= 3:      # (skip the drop of the start pose onto the floor)
        traj.append((before, o.qpos.copy()))
out = {}
for ulps in sys.argv[1:] or ['0', '16']:
  os.environ['DMC_LS_SLOPE_ULPS'] = ulps
  cnt = (ctypes.c_longlong * 6)()
  lib().emu_ls_counts_get(cnt); c0 = list(cnt)
  e = EmuPhysics(m, 32, **caps)
  errs, iters = [], 0
  for (q, v, w, c, a), qn in traj:
    e.qpos[:] = q; e.qvel[:] = v; e.qacc_warmstart[:] = w; e.ctrl[:] = c
    if m.na: e.act[:] = a
    e.step(1, legacy=False)
    errs.append(np.abs(e.qpos - qn).max() / max(1.0, np.abs(qn).max()))
    iters += int(e.solver_iter[0])
  lib().emu_ls_counts_get(cnt)
  errs = np.array(errs)
  out[ulps] = dict(n=len(errs), median=float(np.median(errs)), p90=float(np.percentile(errs, 90)), p99=float(np.percentile(errs, 99)),
                   max=float(errs.max()), evals_per_search=(cnt[1] - c0[1]) / max(1, cnt[0] - c0[0]), iters_per_step=iters / len(errs))
  print('config', cid, 'ulps', ulps, {k: ('%.3g' % x if isinstance(x, float) else x) for k, x in out[ulps].items()}, flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'ls_floor_probe_cfg%d.json' % cid), 'w'), indent=1)
