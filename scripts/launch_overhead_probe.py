"""Where does a single-step launch spend its time?  cheetah B=4096 fp32: t(nstep) per launch, the launch floor
(every environment skipped by env_mode = 2), mj_forward, the kinematic stash on / off, and the wave trace (start ramp
and per-wave duration on the 100 MHz clock)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc  # noqa
from dm_control_amd.batch import BatchedPhysics, OUT  # noqa

name = os.environ.get('MODEL', 'cheetah')
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/%s.xml' % name)).read())
B = int(os.environ.get('B', 4096))
lim = m.jnt_limited == 1
lo, hi = m.jnt_range[lim].T
q0 = np.tile(m.qpos0, (B, 1))
for e in range(B):
  q0[e, lim] = np.random.RandomState(e).uniform(lo, hi)
rs = np.random.RandomState(5)
res = {}


def make(**env):
  for k, v in env.items():
    os.environ[k] = v
  b = BatchedPhysics(m, B, precision=32)
  for k in env:
    del os.environ[k]
  b.set('qpos', q0)
  b.set_output_mask(OUT['sensor'])
  b.step(200); b.sync()
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  return b


b = make()
res['info'] = b.info()
for n in (1, 2, 3, 5, 10):
  res['ms_nstep_%d' % n] = min(b.time_steps(n, 100) for _ in range(3))
b.wave_trace(True)
for _ in range(8):
  b.step()
b.sync()
tr = b.wave_trace().astype(np.int64)
P = lambda x: np.percentile(x, [0, 10, 50, 90, 100]).tolist()
launches = []
for k in range(8):
  ent, st, en = tr[k, 0], tr[k, 1], tr[k, 2]
  t0 = ent.min()
  launches.append(dict(entry=P(ent - t0), start=P(st - t0), end=P(en - t0), dur=P(en - st),
                       gap_from_prev_last_end=float(t0 - tr[k - 1, 2].max()) if k else None,
                       period=float(t0 - tr[k - 1, 0].min()) if k else None))
res['trace'] = dict(launches=launches, unit='10 ns ticks')
b.wave_trace(False)
res['ms_forward'] = None
import time
b.forward(); b.sync()
t = time.perf_counter()
for _ in range(200):
  b.forward()
b.sync()
res['ms_forward'] = (time.perf_counter() - t) / 200 * 1e3
b.set('env_mode', np.full((B, 1), 2, np.int32))
res['ms_floor_env_mode_2'] = min(b.time_steps(1, 200) for _ in range(3))
b.close()
b = make(DMC_NO_KSTASH='1')
res['ms_nokstash_nstep_1'] = min(b.time_steps(1, 100) for _ in range(3))
b.close()
b = make(DMC_STASH='1')
res['ms_fullstash_nstep_1'] = min(b.time_steps(1, 100) for _ in range(3))
b.close()
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'launch_overhead_%s.json' % os.environ.get('TAG', name)), 'w'), indent=1)
