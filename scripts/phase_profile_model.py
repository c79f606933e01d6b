"""Per-phase cycle counts (profiling build of the kernel) for any asset model: MODEL=<asset> NSUB=<substeps>."""
import json, os, sys
os.environ['DMC_USE_PROF'] = '1'
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
name, nsub = os.environ.get('MODEL', 'soccer_2v2_boxhead'), int(os.environ.get('NSUB', 5))
m = mc.compile_xml(common.read_model(name + '.xml'))
B = int(os.environ.get('B', 1024))
rs = np.random.RandomState(5)
q0 = np.tile(m.qpos0, (B, 1))
if name.startswith('soccer'):
  q0[:, [0, 1, 6, 7, 12, 13, 18, 19]] += rs.uniform(-8, 8, (B, 8))
caps = dict(common.DEFAULT_CAPS.get(name, {}))
caps.setdefault('precision', 32)
b = BatchedPhysics(m, B, **caps)
b.set('qpos', q0); b.set_output_mask(OUT['sensor'] | OUT['xpos'] | OUT['xmat'])
for t in range(20):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
b.sync()
print(b.info())
print('mean ncon', b.get('ncon').mean(), 'nefc mean', b.get('nefc').mean(), 'max', b.get('nefc').max(), 'iter', b.get('solver_iter').mean(), b.get('warning').sum(axis=0))
b.prof_enable(True)
N = 5
for t in range(N):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
p = b.prof_get()
tot = sum(p.values())
print('%s: total cycles per physics step %.0f' % (name, tot / N / nsub))
out = {}
for k, v in sorted(p.items(), key=lambda kv: -kv[1]):
  if v:
    print('   %-16s %9.0f  %5.1f%%' % (k, v / N / nsub, 100 * v / tot))
    out[k] = v / N / nsub
json.dump(dict(model=name, cycles_per_physics_step=out, info=b.info()), open(os.path.join(ROOT, 'gpurun_out', 'phase_%s.json' % name), 'w'), indent=1)
