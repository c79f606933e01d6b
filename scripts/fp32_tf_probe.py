"""CPU study of the fp32 teacher-forced error on a BASELINE model: the host build of the kernel core (tests/emu, fp32,
exact 1/sqrt) against the fp64 oracle, one env-step at a time from the oracle's state.  Finds the worst steps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from emu_lib import EmuPhysics
from oracle import oracle
from dm_control_amd.suite import common

cfgid = int(os.environ.get('CFG', '4')); NE = int(os.environ.get('NE', '8')); T = int(os.environ.get('T', '40'))
prec = int(os.environ.get('PREC', '32'))
cfg = bench.CONFIGS[cfgid]; nsub = cfg['nsub']
per_physics_step = bool(os.environ.get('PHYS'))
if per_physics_step:
  T, nsub = T * nsub, 1      # forced every physics step (legacy Physics.step(1))
m = bench.load_model(cfg['asset'])
caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {})); caps.pop('precision', None)
q0 = bench.initial_qpos(cfg, m, NE, seed0=0)
om = oracle.OracleModel(m)
rs = np.random.RandomState(77)
worst = []
for e in range(NE):
  p = oracle.OraclePhysics(om); p.qpos[:] = q0[e]; p.forward()
  g = EmuPhysics(m, prec=prec, **{k: v for k, v in caps.items() if k in ('nconmax', 'njmax', 'njcon')})
  for t in range(T):
    a = rs.uniform(-1, 1, m.nu).astype(np.float32).astype(np.float64)
    g.qpos[:] = p.qpos; g.qvel[:] = p.qvel; g.qacc_warmstart[:] = p.qacc_warmstart; g.time[:] = p.time
    g.ctrl[:] = a; p.set_control(a)
    g.step(nsub); p.step(nsub)
    err = np.abs(g.qpos - p.qpos).max() / max(1.0, np.abs(p.qpos).max())
    worst.append((err, e, t, int(p.ncon), int(p.nefc), int(p.solver_iter), int(g.ncon[0]), int(g.solver_iter[0])))
worst.sort(reverse=True)
errs = np.array([w[0] for w in worst])
print('cfg %d prec %d: %d env-steps; median %.2e p90 %.2e p99 %.2e max %.2e' % (cfgid, prec, len(errs), np.median(errs), np.percentile(errs, 90), np.percentile(errs, 99), errs.max()))
for w in worst[:12]:
  print('  err %.2e env %d step %d oracle ncon %d nefc %d iter %d | emu ncon %d iter %d' % w)
