"""Wave durations of single-step launches under the bench workload (random actions every step): how much of a launch
is its slowest wave, and what makes that wave slow (solver iterations, contacts)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
name = os.environ.get('MODEL', 'cheetah')
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/%s.xml' % name)).read())
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
out = []
epw = 64 // b.info()['lanes_per_env']
for rep in range(4):
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  for _ in range(8):
    b.step()
  b.sync()
  tr = b.wave_trace().astype(np.int64)
  it = b.get('solver_iter')[:, 0].reshape(-1, epw).max(axis=1)
  nc = b.get('ncon')[:, 0].reshape(-1, epw).max(axis=1)
  k = 7
  ent0 = tr[k, 0].min(); dur = tr[k, 2] - tr[k, 1]
  seg = lambda a, b: (tr[k, b] - tr[k, a]).astype(np.float64)
  stages = dict(staging=seg(0, 1), posvel_open=tr[k, 4] - tr[k, 1], acc=seg(4, 5), euler=seg(5, 6), posvel_trailing=seg(6, 7), store=tr[k, 2] - tr[k, 7])
  rec = dict(stage_mean_by_iter={name: {int(v): float(np.mean(x[it == v])) for v in np.unique(it)} for name, x in stages.items()},
             span=int(tr[k, 2].max() - ent0), period=int(ent0 - tr[k - 1, 0].min()), dur_pct=np.percentile(dur, [0, 10, 50, 90, 99, 100]).tolist(),
             by_iter={int(v): [int((it == v).sum()), float(dur[it == v].mean()), int(dur[it == v].max())] for v in np.unique(it)},
             by_ncon={int(v): [int((nc == v).sum()), float(dur[nc == v].mean()), int(dur[nc == v].max())] for v in np.unique(nc)})
  out.append(rec)
  print(json.dumps(rec))
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'tail_probe_%s.json' % name), 'w'), indent=1)
