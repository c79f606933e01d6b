"""Does overlapping two half-batches on two HIP streams hide the per-launch tail?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/cheetah.xml')).read())
lim = m.jnt_limited == 1
lo, hi = m.jnt_range[lim].T
rs = np.random.RandomState(0)
def make(B):
  b = BatchedPhysics(m, B, precision=32)
  q = np.tile(m.qpos0, (B, 1)); q[:, lim] = rs.uniform(lo, hi, (B, lo.size))
  b.set('qpos', q); b.set_output_mask(OUT['sensor']); b.step(200); b.sync()
  return b
K = 500
for nsplit in (1, 2, 4):
  Bs = 4096 // nsplit
  parts = [make(Bs) for _ in range(nsplit)]
  streams = [torch.cuda.Stream() for _ in range(nsplit)]
  acts = [(torch.rand((K, m.nu, Bs), device='cuda') * 2 - 1) for _ in range(nsplit)]
  for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(K):
      for p, s, a in zip(parts, streams, acts):
        p.bind('ctrl', a[t].data_ptr())
        p.step(1, stream=s.cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print('nsplit', nsplit, 'ms/step %.4f' % (dt / K * 1e3), 'Msteps/s %.2f' % (4096 * K / dt / 1e6), flush=True)
  for p in parts: p.close()
