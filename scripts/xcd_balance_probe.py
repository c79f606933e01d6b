"""Per-XCD finish times of a sliced config-3 / config-4 launch (CONFIG=3|4): the queues are per XCD (step_kernel_body), so a
launch ends with its slowest XCD.  Reads the wave trace: end of every item's last piece and the workgroup that ran it
(workgroup b runs on XCD b % 8: observed dispatch order)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
cfgid = int(os.environ.get('CONFIG', 4))
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
b.wave_trace(True)
out = []
for rep in range(4):
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  for _ in range(8): b.step(nsub)
  b.sync()
  tr = b.wave_trace().astype(np.int64)
  k = 7
  ent0 = tr[k, 0].min()
  end, blk = tr[k, 2] - ent0, tr[k, 3]
  xcd = blk & 7
  last = [int(end[xcd == x].max()) for x in range(8)]
  cnt = [int((xcd == x).sum()) for x in range(8)]
  out.append(dict(span=int(end.max()), xcd_last_end=last, items_finished_per_xcd=cnt, spread=float((max(last) - min(last)) / max(last)), mean_over_max=float(np.mean(last) / max(last))))
print(json.dumps(out))
