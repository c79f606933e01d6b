"""Wave durations of one launch per env-step under the bench workload of a BASELINE config (CONFIG=3..5): how much of a
launch is its slowest wave, what makes it slow, and (NOSLIP0=1) what the launch costs without the noslip pass."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
cfgid = int(os.environ.get('CONFIG', 5))
cfg = bench.CONFIGS[cfgid]
m = mc.compile_xml(common.read_model(cfg['asset'] + '.xml'))
B = int(os.environ.get('B', cfg['batch']))
caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {})); caps.pop('precision', None)
b = BatchedPhysics(m, B, precision=32, **caps)
b.set('qpos', bench.initial_qpos(cfg, m, B, 0, phys=b))
mask = 0
for n in cfg['outputs']: mask |= OUT[n]
b.set_output_mask(mask)
rs = np.random.RandomState(5)
nsub = cfg['nsub']
b.forward(); b.sync()
for t in range(100):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
b.sync()
res = dict(config=cfgid, info=b.info())
res['ms_launch'] = min(b.time_steps(nsub, 20) for _ in range(3))
if os.environ.get('NOSLIP0'):
  b.set_opt('noslip_iterations', 0)
  res['ms_launch_noslip0'] = min(b.time_steps(nsub, 20) for _ in range(3))
  b.set_opt('noslip_iterations', 5)
b.wave_trace(True)
epw = 64 // b.info()['lanes_per_env']
out = []
for rep in range(3):
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  for _ in range(8):
    b.step(nsub)
  b.sync()
  tr = b.wave_trace().astype(np.int64)
  it = b.get('solver_iter')[:, 0].reshape(-1, epw).max(axis=1)
  nc = b.get('ncon')[:, 0].reshape(-1, epw).max(axis=1)
  k = 7
  ent0 = tr[k, 0].min(); dur = tr[k, 2] - tr[k, 1]
  out.append(dict(span=int(tr[k, 2].max() - ent0), period=int(ent0 - tr[k - 1, 0].min()), dur_pct=np.percentile(dur, [0, 10, 50, 90, 99, 100]).tolist(),
                  by_ncon={int(v): [int((nc == v).sum()), float(dur[nc == v].mean()), int(dur[nc == v].max())] for v in np.unique(nc)},
                  by_iter_last_substep={int(v): [int((it == v).sum()), float(dur[it == v].mean())] for v in np.unique(it)}))
res['trace'] = out
print(json.dumps(res))
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'tail_probe_cfg%d.json' % cfgid), 'w'), indent=1)
