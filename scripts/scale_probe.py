import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/cheetah.xml')).read())
lim = m.jnt_limited == 1
lo, hi = m.jnt_range[lim].T
rs = np.random.RandomState(5)
for lanes in (64, 32, 16):
  for B in (16, 256, 1024, 2048, 4096, 8192, 16384, 65536):
    q0 = np.tile(m.qpos0, (B, 1))
    q0[:, lim] = rs.uniform(lo, hi, (B, lim.sum()))
    b = BatchedPhysics(m, B, precision=32, lanes_per_env=lanes)
    b.set('qpos', q0); b.set_output_mask(OUT['sensor'])
    b.step(200); b.sync()
    b.set_control(rs.uniform(-1, 1, (B, m.nu)))
    ms = b.time_steps(1, 30)
    i = b.info()
    print('lanes %2d B %6d  ms %.4f  Msteps/s %7.2f  grid %5d epb %2d lds %6d' % (lanes, B, ms, B / ms / 1e3, i['grid'], i['envs_per_block'], i['lds_bytes_per_block']), flush=True)
    b.close()
