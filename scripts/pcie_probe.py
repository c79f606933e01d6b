"""PCIe-inclusive rate of the HOST-buffer side of the boundary (dmc_batch_set_real / dmc_batch_get_real): per env-step
the controls come from host memory and qpos / qvel / sensordata go back to host memory, as a caller without device
tensors would use the library.  Never the bench `value` (inputs resident in HBM); DESIGN.md section 5 quotes it.

  CONFIG=2 python scripts/pcie_probe.py   -> gpurun_out/pcie_probe_cfg2.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc  # noqa: E402
from dm_control_amd.batch import BatchedPhysics, OUT  # noqa: E402
from dm_control_amd.suite import common  # noqa: E402

cfg = int(os.environ.get('CONFIG', 2))
asset, B, nsub, caps = {2: ('cheetah', 4096, 1, {}), 3: ('humanoid', 4096, 5, dict(nconmax=24))}[cfg]
m = mc.compile_xml(common.read_model(asset + '.xml'))
b = BatchedPhysics(m, B, precision=32, **caps)
b.set_output_mask(OUT['sensor'])
rs = np.random.RandomState(0)
b.step(50); b.sync()
T = int(os.environ.get('T', 200))
ctrl = rs.uniform(-1, 1, (T, B, m.nu))
for t in range(10):
  b.set('ctrl', ctrl[t]); b.step(nsub); b.get('qpos'); b.get('qvel'); b.get('sensordata')
t0 = time.perf_counter()
for t in range(T):
  b.set('ctrl', ctrl[t])
  b.step(nsub)
  q, v, s = b.get('qpos'), b.get('qvel'), b.get('sensordata')
dt = time.perf_counter() - t0
# the asynchronous boundary (dmc_batch_set_async / get_async / get_wait): controls staged through pinned memory, the three
# fields fetched with ONE device-to-host copy and one wait per env-step; fp64 host arrays, then fp32 on the wire
rates = {}
for tag, hdt in (('f64_host', np.float64), ('f32_host', np.float32)):
  c = ctrl.astype(hdt)
  for t in range(10):
    b.set_async('ctrl', c[t]); b.step(nsub); b.get_many(('qpos', 'qvel', 'sensordata'), dtype=hdt)
  t1 = time.perf_counter()
  for t in range(T):
    b.set_async('ctrl', c[t])
    b.step(nsub)
    o = b.get_many(('qpos', 'qvel', 'sensordata'), dtype=hdt)
  rates[tag] = B * T / (time.perf_counter() - t1)
t1 = time.perf_counter()
c = ctrl.astype(np.float32)
for t in range(T):
  b.set_async('ctrl', c[t])
  b.step(nsub)
  o = b.get_many(('qpos', 'qvel', 'sensordata'), copy=False)      # views of the pinned staging: no host copy
rates['f32_host_zero_copy'] = B * T / (time.perf_counter() - t1)
# and with the observation download of step t overlapped with the launch of step t + 1 (the policy sees a one-step-old
# observation only if it wants to; here the next action does not depend on it: random actions)
c = ctrl.astype(np.float32)
b.set_async('ctrl', c[0]); b.step(nsub); b.get_async(('qpos', 'qvel', 'sensordata'))
t1 = time.perf_counter()
for t in range(1, T):
  b.set_async('ctrl', c[t])
  b.step(nsub)
  o = b.get_wait(np.float32)
  b.get_async(('qpos', 'qvel', 'sensordata'))
b.get_wait(np.float32)
rates['f32_host_pipelined'] = B * (T - 1) / (time.perf_counter() - t1)
# the numpy FACADE a drop-in user gets (Physics.set_control / step / data.qpos ...): reads are fetched together from the second step on
from dm_control_amd import physics as physics_lib  # noqa: E402
fp = physics_lib.Physics(m, batch_size=B, precision=32, **caps)
fp.step(50)
for t in range(10):
  fp.set_control(ctrl[t]); fp.step(nsub); _ = (fp.data.qpos, fp.data.qvel, fp.data.sensordata)
t1 = time.perf_counter()
for t in range(T):
  fp.set_control(ctrl[t]); fp.step(nsub)
  _ = (fp.data.qpos, fp.data.qvel, fp.data.sensordata)
rates['numpy_facade'] = B * T / (time.perf_counter() - t1)
fp.free()
ms_dev = b.time_steps(nsub, 100)
out = dict(config=cfg, B=B, env_steps=T, host_buffer_env_steps_per_s=B * T / dt, ms_per_env_step_host_buffers=1e3 * dt / T,
           ms_per_launch_device_resident=ms_dev, device_resident_env_steps_per_s=B / (ms_dev * 1e-3),
           async_boundary_env_steps_per_s=rates,
           bytes_per_env_step=dict(host_to_device=8 * B * m.nu, device_to_host=8 * B * (m.nq + m.nv + m.nsensordata)),
           note='fp64 host arrays (the facade\'s numpy dtype) converted to / from the fp32 SoA device fields by the library; '
                'one synchronous set + 3 synchronous gets per env-step')
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'pcie_probe_cfg%d.json' % cfg), 'w') as f:
  json.dump(out, f, indent=1)
print(json.dumps(out))
b.close()
