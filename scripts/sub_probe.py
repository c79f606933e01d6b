"""Investigation builds (-DDMC_TRACE_SUB=<region>, library variants sub1 / sub2 / sub3): rows 4..7 of the wave trace are
stamps INSIDE one region of the pipeline.  Prints the mean segment lengths (us) over the waves of a single-step cheetah
launch under the bench workload, by the Newton iteration count of the wave."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
# REGION=<n>: the on-demand specialised kernel built with DMC_SPEC_FLAGS=-DDMC_TRACE_SUB=<n> (scripts/spec_variants.py);
# B=<batch>: 512 = one wave per CU, i.e. the latency of a wave that has its SIMD to itself (the tail of a launch)
if os.environ.get('REGION'):
  region = int(os.environ['REGION'])
  os.environ['DMC_NO_STATIC'] = '1'
  os.environ['DMC_SPEC_FLAGS'] = (os.environ.get('EXTRA_FLAGS', '') + ' -DDMC_TRACE_SUB=%d' % region).strip()
else:
  region = int(os.environ['DMC_LIB_VARIANT'][3:])
names = {1: ['start->after kinematics (acc, euler, kin)', 'com_pos', 'sensors(pos)', 'com_vel', 'sensors(vel)+store'],
         2: ['start->after crb+factor', 'collision', 'make_constraint', 'sensors(pos)+rne', 'rest (acc, euler, trailing, store)'],
         3: ['start->first iteration', 'primal_search', 'update+constraint_update+gauss', 'newton_gradient', 'rest']}[region]
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/cheetah.xml')).read())
B = int(os.environ.get('B', 4096))
lim = m.jnt_limited == 1
lo, hi = m.jnt_range[lim].T
q0 = np.tile(m.qpos0, (B, 1))
for e in range(B):
  q0[e, lim] = np.random.RandomState(e).uniform(lo, hi)
rs = np.random.RandomState(5)
b = BatchedPhysics(m, B, precision=32)
b.set('qpos', q0); b.set_output_mask(OUT['sensor']); b.step(200); b.sync()
for t in range(300):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step()
b.sync()
b.wave_trace(True)
b.set_control(rs.uniform(-1, 1, (B, m.nu)))
for _ in range(8):
  b.step()
b.sync()
tr = b.wave_trace().astype(np.int64)
it = b.get('solver_iter')[:, 0].reshape(-1, 2).max(axis=1)
k = 7
pts = [tr[k, 1], tr[k, 4], tr[k, 5], tr[k, 6], tr[k, 7], tr[k, 2]]
ok = np.all([p > 0 for p in pts], axis=0)
if region == 3:
  ok &= it >= 1
print('region', region, 'B', B, 'waves with all stamps', int(ok.sum()), b.info().get('static_id'))
for i, n in enumerate(names):
  seg = (pts[i + 1] - pts[i])[ok] / 100.0
  by = {int(v): round(float(seg[it[ok] == v].mean()), 2) for v in np.unique(it[ok])}
  print('  %-48s mean %6.2f us   by iterations %s' % (n, seg.mean(), by))
