"""What a SIXTH resident 62-dof environment per CU would be worth (config 4's workload): the contact caps lowered until six
environments fit a CU's LDS (nconmax 16: 24.1 KB per environment; the bench workload has 2.4 contacts on average, 26 at most,
so a handful of environments drop contacts -- the timing is that of the same work), stepped with five and with six waves
per workgroup (DMC_WAVES; kernels built for 384 threads).  WAVES=5|6 python scripts/residency_probe.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W = int(os.environ.get('WAVES', 5))
os.environ['DMC_WAVES'] = str(W)
os.environ['DMC_NO_STATIC'] = '1'; os.environ['DMC_SPECIALISE'] = 'build'
os.environ['DMC_SPEC_FLAGS'] = '-DDMC_MAX_THREADS=384'
import numpy as np
import bench
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
cfg = bench.CONFIGS[4]
m = mc.compile_xml(common.read_model(cfg['asset'] + '.xml'))
B = cfg['batch']
b = BatchedPhysics(m, B, precision=32, nconmax=int(os.environ.get('NCONMAX', 16)))
b.set('qpos', bench.initial_qpos(cfg, m, B, 0, phys=b))
mask = 0
for n in cfg['outputs']: mask |= OUT[n]
b.set_output_mask(mask)
rs = np.random.RandomState(5)
nsub = cfg['nsub']
b.forward(); b.sync()
for t in range(60):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
b.sync()
info = b.info()
ms = min(b.time_steps(nsub, 20) for _ in range(3))
print(json.dumps(dict(waves=W, envs_per_cu=info['envs_per_cu'], lds_bytes_per_block=info['lds_bytes_per_block'], env_scratch_bytes=info['env_scratch_bytes'],
                      ms_per_launch=ms, env_steps_per_s=B / ms * 1e3, mean_ncon=float(b.get('ncon').mean()), max_ncon=int(b.get('ncon').max()),
                      warnings=b.get('warning').sum(axis=0).tolist(), static_id=info.get('static_id'))))
