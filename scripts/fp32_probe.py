"""fp32 one-env-step error of the manipulator from the task's own (collision-free) start states."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from dm_control_amd import mjcf_compiler as mc, suite
from dm_control_amd.batch import BatchedPhysics
from oracle import oracle
from oracle.oracle import OraclePhysics
NE = 16
env = suite.load('manipulator', 'insert_ball', task_kwargs=dict(random=5), physics_kwargs=dict(batch_size=NE))
env.reset()
m = env.physics.model
q0 = np.array(env.physics.data.qpos); v0 = np.array(env.physics.data.qvel)
for prec in (32, 64):
  refs = []
  for e in range(NE):
    o = OraclePhysics(m); o.qpos[:] = q0[e]; o.qvel[:] = v0[e]; o.forward(); refs.append(o)
  b = BatchedPhysics(m, NE, precision=prec)
  rs = np.random.RandomState(3); errs = []
  for t in range(100):
    a = rs.uniform(-1, 1, (NE, m.nu))
    b.set('qpos', np.stack([o.qpos for o in refs])); b.set('qvel', np.stack([o.qvel for o in refs])); b.set('qacc_warmstart', np.stack([o.qacc_warmstart for o in refs]))
    b.set_control(a); b.step(10)
    oracle.rollout_legacy(refs, a[None], nsub=10)
    qo = np.stack([o.qpos for o in refs])
    errs.append(np.abs(b.get('qpos') - qo).max(axis=1) / np.maximum(1.0, np.abs(qo).max(axis=1)))
  errs = np.array(errs)
  print(prec, 'max %.2e median %.2e frac<=5e-5 %.3f' % (errs.max(), np.median(errs), (errs <= 5e-5).mean()), 'maxcon', int(b.get('ncon').max()), 'iters', int(b.get('solver_iter').max()), flush=True)
  w = np.argwhere(errs > 5e-5)
  print('  offenders (step, env):', w[:10].tolist())
  b.close()
