import json, os, sys
os.environ['DMC_USE_PROF'] = '1'
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/humanoid.xml')).read())
B = 2048
rs = np.random.RandomState(5)
q0 = np.tile(m.qpos0, (B, 1)); q0[:, 7:] += rs.uniform(-0.2, 0.2, (B, m.nq - 7))
from dm_control_amd.suite import common
b = BatchedPhysics(m, B, precision=32, lanes_per_env=64, **common.DEFAULT_CAPS['humanoid'])
b.set('qpos', q0); b.set_output_mask(OUT['sensor'] | OUT['xpos'] | OUT['xmat'] | OUT['subtree_com'])
for t in range(60):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(5)
b.sync()
print('mean ncon', b.get('ncon').mean(), 'max', b.get('ncon').max(), 'nefc mean', b.get('nefc').mean(), 'max', b.get('nefc').max(), 'iter', b.get('solver_iter').mean(), b.get('solver_iter').max(), b.get('warning').sum(axis=0))
b.prof_enable(True)
N = 10
for t in range(N):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(5)
p = b.prof_get()
tot = sum(p.values())
print('humanoid f32 lanes64: total cycles per env-step(5 substeps) %.0f' % (tot / N))
for k, v in sorted(p.items(), key=lambda kv: -kv[1]):
  if v: print('   %-16s %9.0f  %5.1f%%' % (k, v / N / 5, 100 * v / tot))
