import sys, time, numpy as np
sys.path.insert(0, '.')
from dm_control_amd import mjcf_compiler as mc, physics as pl
from dm_control_amd.suite import common
m = mc.compile_xml(common.read_model('cheetah.xml'))
B = 4096
fp = pl.Physics(m, batch_size=B, precision=32)
rs = np.random.RandomState(0); ctrl = rs.uniform(-1, 1, (60, B, m.nu))
fp.step(50)
for t in range(10):
  fp.set_control(ctrl[t]); fp.step(); _ = (fp.data.qpos, fp.data.qvel, fp.data.sensordata)
t1 = time.perf_counter()
for t in range(200):
  fp.set_control(ctrl[t % 60]); fp.step(); _ = (fp.data.qpos, fp.data.qvel, fp.data.sensordata)
print('numpy facade env-steps/s', B * 200 / (time.perf_counter() - t1))
