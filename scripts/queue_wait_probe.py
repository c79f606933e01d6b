"""Where the wave slots of a sliced launch spend the launch (CONFIG=3|4): per item, the ticks its pieces ran and the ticks they
waited for their predecessor piece (wave trace rows 5 / 4), against slots x span.  Kernel built on the spot (the trace rows
are written by step_kernel.hip.h)."""
import json, os, sys
os.environ['DMC_NO_STATIC'] = '1'; os.environ['DMC_SPECIALISE'] = 'build'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
cfgid = int(os.environ.get('CONFIG', 3))
cfg = bench.CONFIGS[cfgid]
m = mc.compile_xml(common.read_model(cfg['asset'] + '.xml'))
B = cfg['batch']
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
info = b.info()
ms = min(b.time_steps(nsub, 20) for _ in range(3))
b.wave_trace(True)
out = []
for rep in range(3):
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  for _ in range(8):
    b.step(nsub)
  b.sync()
  tr = b.wave_trace().astype(np.int64)
  k = 7
  ent0 = tr[k, 0].min(); span = int(tr[k, 2].max() - ent0)
  slots = info['envs_per_cu'] * 256 * info['lanes_per_env'] // 64
  wait, run = tr[k, 4], tr[k, 5]
  out.append(dict(span_ticks=span, slots=slots, items=int(tr.shape[2]), run_frac=float(run.sum() / (slots * span)), wait_frac=float(wait.sum() / (slots * span)),
                  wait_per_item_pct=np.percentile(wait, [0, 50, 90, 99, 100]).tolist(), run_per_item_pct=np.percentile(run, [0, 50, 90, 99, 100]).tolist()))
res = dict(config=cfgid, ms_launch=ms, envs_per_cu=info['envs_per_cu'], static_id=info.get('static_id'), trace=out)
print(json.dumps(res))
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'queue_wait_probe_cfg%d.json' % cfgid), 'w'), indent=1)
