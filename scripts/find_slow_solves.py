import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/cheetah.xml')).read())
B = 4096
lim = m.jnt_limited == 1
lo, hi = m.jnt_range[lim].T
q0 = np.tile(m.qpos0, (B, 1))
for e in range(B):
  q0[e, lim] = np.random.RandomState(e).uniform(lo, hi)
rs = np.random.RandomState(5)
lanes = int(os.environ.get('LANES', 64))
b = BatchedPhysics(m, B, precision=32, lanes_per_env=lanes)
b.set('qpos', q0)
b.step(200)
saved = []
hist = np.zeros(101, int)
for t in range(300):
  c = rs.uniform(-1, 1, (B, m.nu))
  b.set_control(c)
  pre = (b.get('qpos'), b.get('qvel'), b.get('qacc_warmstart'))
  b.step()
  it = b.get('solver_iter')[:, 0]
  hist += np.bincount(np.minimum(it, 100), minlength=101)
  for e in np.nonzero(it >= 20)[0][:4]:
    if len(saved) < 40:
      saved.append(dict(t=t, env=int(e), it=int(it[e]), nefc=int(b.get('nefc')[e, 0]), qpos=pre[0][e].tolist(),
                        qvel=pre[1][e].tolist(), warm=pre[2][e].tolist(), ctrl=c[e].tolist()))
print('iter hist', {i: int(c) for i, c in enumerate(hist) if c})
json.dump(saved, open(os.path.join(ROOT, 'gpurun_out', 'slow_solves.json'), 'w'))
print(len(saved), [(s['t'], s['env'], s['it'], s['nefc']) for s in saved[:10]])
