"""Newton iteration counts of a BASELINE config under the bench workload, per PHYSICS step (single-substep launches so
that the state before the offending solve is known), and the states of the environments whose solve ran long -- saved for
a teacher-forced replay on the host (tests/emu, oracle): CONFIG=4 [B=4096] [STEPS=40] [LONG=40] python scripts/iter_hist.py
-> gpurun_out/iter_hist_cfg<N>.json"""
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
B = int(os.environ.get('B', cfg['batch']))
T = int(os.environ.get('STEPS', 40))
LONG = int(os.environ.get('LONG', 40))
caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {})); caps.pop('precision', None)
b = BatchedPhysics(m, B, precision=32, **caps)
b.set('qpos', bench.initial_qpos(cfg, m, B, 0, phys=b))
rs = np.random.RandomState(5)
nsub = cfg['nsub']
b.forward(); b.sync()
for t in range(int(os.environ.get('SETTLE', 60))):      # the bench's warm-up: ragdolls on the floor, players moving
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
hist = np.zeros(m.opt.iterations + 2, dtype=np.int64)
saved = []
ncon_hist = {}
for t in range(T):
  a = rs.uniform(-1, 1, (B, m.nu))
  b.set_control(a)
  for k in range(nsub):
    pre = b.get_many(('qpos', 'qvel', 'qacc_warmstart') + (('act',) if m.na else ()))
    b.step(1)
    it = b.get('solver_iter')[:, 0]
    hist += np.bincount(np.minimum(it, len(hist) - 1), minlength=len(hist))
    for e in np.nonzero(it >= LONG)[0][:4]:
      if len(saved) < 16:
        saved.append(dict(env=int(e), step=t, substep=k, iters=int(it[e]), ncon=int(b.get('ncon')[e, 0]), nefc=int(b.get('nefc')[e, 0]),
                          ctrl=a[e].tolist(), **{n: pre[n][e].tolist() for n in pre}))
tot = int(hist.sum())
nz = {int(i): int(c) for i, c in enumerate(hist) if c}
cum = np.cumsum(hist) / tot
out = dict(config=cfgid, B=B, physics_steps=tot, histogram=nz, mean=float((np.arange(len(hist)) * hist).sum() / tot),
           p50=int(np.searchsorted(cum, 0.5)), p99=int(np.searchsorted(cum, 0.99)), p999=int(np.searchsorted(cum, 0.999)),
           frac_ge_20=float(hist[20:].sum() / tot), frac_at_cap=float(hist[m.opt.iterations:].sum() / tot), cap=int(m.opt.iterations),
           warnings=[int(w) for w in b.get('warning').sum(axis=0)], long_solves=saved)
print(json.dumps({k: v for k, v in out.items() if k != 'long_solves'}))
print('saved', len(saved), 'long solves')
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'iter_hist_cfg%d.json' % cfgid), 'w'))
