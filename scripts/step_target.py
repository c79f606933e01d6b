"""Steps one asset model a few times (profiling target for rocprofv3): MODEL=<asset> NSUB=<substeps> B=<batch> REPS=<launches>."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
name, nsub = os.environ.get('MODEL', 'cheetah'), int(os.environ.get('NSUB', 1))
B, reps = int(os.environ.get('B', 4096)), int(os.environ.get('REPS', 30))
m = mc.compile_xml(common.read_model(name + '.xml'))
rs = np.random.RandomState(5)
q0 = np.tile(m.qpos0, (B, 1))
if name.startswith('soccer'):
  q0[:, [0, 1, 6, 7, 12, 13, 18, 19]] += rs.uniform(-8, 8, (B, 8))
elif name == 'cheetah':
  lim = m.jnt_limited == 1
  lo, hi = m.jnt_range[lim].T
  q0[:, lim] = rs.uniform(lo, hi, (B, lo.size))
caps = dict(common.DEFAULT_CAPS.get(name, {}))
caps.setdefault('precision', 32)
b = BatchedPhysics(m, B, **caps)
b.set('qpos', q0); b.set_output_mask(OUT['sensor'] | OUT['xpos'] | OUT['xmat'])
for t in range(20 if name != 'cheetah' else 200):
  b.set_control(rs.uniform(-1, 1, (B, m.nu)) * (name != 'cheetah' or t > 190)); b.step(nsub)
b.sync()
for t in range(reps):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
b.sync()
print(b.info(), 'ncon', b.get('ncon').mean(), 'nefc', b.get('nefc').mean(), 'iter', b.get('solver_iter').mean())
