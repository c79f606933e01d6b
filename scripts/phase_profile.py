"""Per-phase shader-cycle breakdown of the fused step kernel (DMC_PROFILE build)."""
import json, os, sys
os.environ['DMC_USE_PROF'] = '1'
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/cheetah.xml')).read())
B = 4096
lim = m.jnt_limited == 1
lo, hi = m.jnt_range[lim].T
q0 = np.tile(m.qpos0, (B, 1))
for e in range(B):
  q0[e, lim] = np.random.RandomState(e).uniform(lo, hi)
rs = np.random.RandomState(5)
res = {}
for prec, lanes in ((32, 32),):
  b = BatchedPhysics(m, B, precision=prec, lanes_per_env=lanes)
  b.set('qpos', q0); b.set_output_mask(OUT['sensor'])
  b.step(200); b.sync()
  b.prof_enable(True)
  N = 50
  for t in range(N):
    b.set_control(rs.uniform(-1, 1, (B, m.nu)))
    b.step()
  p = b.prof_get()
  tot = sum(p.values())
  key = 'f%d_lanes%d' % (prec, lanes)
  res[key] = {k: v / N for k, v in p.items()}
  print(key, 'total cycles/env-step %.0f' % (tot / N))
  for k, v in sorted(p.items(), key=lambda kv: -kv[1]):
    print('   %-16s %9.0f  %5.1f%%' % (k, v / N, 100 * v / tot))
  b.close()
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'phase_profile.json'), 'w'), indent=1)
