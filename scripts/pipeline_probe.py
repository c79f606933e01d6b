"""Does splitting the batch into P independent parts on P streams (part p's launch t+1 only waits for part p's launch t)
hide the tail of a queued launch?  CONFIG = 2..5, PARTS = "1,2,4"."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
cfgid = int(os.environ.get('CONFIG', 4))
cfg = bench.CONFIGS[cfgid]
m = bench.load_model(cfg['asset'])
B = int(os.environ.get('B', cfg['batch']))
K = int(os.environ.get('K', 100))
nsub = cfg['nsub']
caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {})); caps.pop('precision', None)
mask = 0
for n in cfg['outputs']: mask |= OUT[n]
dev = torch.device('cuda', 0)
rs = np.random.RandomState(5)
acts = torch.from_numpy(np.ascontiguousarray(rs.uniform(-1, 1, (K + 20, B, m.nu)).astype(np.float32).transpose(0, 2, 1))).to(dev)
out = dict(config=cfgid, B=B, K=K)
for P in [int(x) for x in os.environ.get('PARTS', '1,2,4').split(',')]:
  Bp = B // P
  parts, streams, ctrl = [], [], []
  for p in range(P):
    b = BatchedPhysics(m, Bp, precision=32, **caps)
    b.set('qpos', bench.initial_qpos(cfg, m, Bp, p * Bp, phys=b))
    b.set_output_mask(mask)
    b.forward()
    parts.append(b); streams.append(torch.cuda.Stream())
    ctrl.append(acts[:, :, p*Bp:(p+1)*Bp].contiguous())
  torch.cuda.synchronize()
  def run(t0, n):
    for t in range(t0, t0 + n):
      for p in range(P):
        parts[p].bind('ctrl', ctrl[p][t].data_ptr())
        parts[p].step(nsub, stream=streams[p].cuda_stream)
  run(0, 20)
  torch.cuda.synchronize()
  best = 1e9
  for rep in range(3):
    t0 = time.perf_counter()
    run(20, K)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
  out['parts_%d' % P] = dict(env_steps_per_s=B * K / best, ms_per_env_step=best / K * 1e3, grid=parts[0].info()['grid'])
  print(P, out['parts_%d' % P], flush=True)
  for b in parts: b.close()
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'pipeline_cfg%d.json' % cfgid), 'w'), indent=1)
