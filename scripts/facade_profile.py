"""cProfile of the numpy facade's host loop (config 2, B = 4096: set_control / step / read qpos, qvel, sensordata): where the
0.3 ms per step go -- the wait for the launch + download (get_wait), the equality checks that spare re-uploading inputs
that were only read, the rest of the Python around a 0.087 ms kernel.  gpurun_out/facade_profile.log."""
import sys, time, numpy as np, cProfile, pstats, io
sys.path.insert(0, '.')
from dm_control_amd import mjcf_compiler as mc, physics as pl
from dm_control_amd.suite import common
m = mc.compile_xml(common.read_model('cheetah.xml'))
B = 4096
fp = pl.Physics(m, batch_size=B, precision=32)
rs = np.random.RandomState(0); ctrl = rs.uniform(-1, 1, (60, B, m.nu))
fp.step(50)
def loop(n):
  for t in range(n):
    fp.set_control(ctrl[t % 60]); fp.step(); _ = (fp.data.qpos, fp.data.qvel, fp.data.sensordata)
loop(10)
t1 = time.perf_counter(); loop(200); print('env-steps/s', B * 200 / (time.perf_counter() - t1))
pr = cProfile.Profile(); pr.enable(); loop(200); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])
