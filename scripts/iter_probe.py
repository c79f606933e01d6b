"""Solver iteration statistics + state checksum of a short cheetah rollout (variant comparison)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics

A = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dm_control_amd', 'suite', 'assets')
m = mc.compile_xml(open(os.path.join(A, 'cheetah.xml')).read())
B = 1024
b = BatchedPhysics(m, B, precision=int(os.environ.get('PREC', '32')), lanes_per_env=32)
rs = np.random.RandomState(0)
q = np.tile(m.qpos0, (B, 1)); q[:, 3:] += rs.uniform(-0.3, 0.3, (B, 6))
b.set('qpos', q)
tot = 0; mx = 0; hist = np.zeros(102, int)
for t in range(200):
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  b.step()
  it = b.get('solver_iter')[:, 0]
  tot += it.sum(); mx = max(mx, it.max()); hist += np.bincount(it, minlength=102)[:102]
print(json.dumps(dict(variant=os.environ.get('DMC_LIB_VARIANT', ''), mean_iter=tot / (200 * B), max_iter=int(mx),
                      hist=hist[:12].tolist(), tail=int(hist[12:].sum()), qsum=float(np.abs(b.get('qpos')).sum()))))
