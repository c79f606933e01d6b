"""Per-phase cycle counts (profiling build, model-specialised kernels included) under the bench workload of a BASELINE
config: CONFIG=2..5 [B=<batch>]."""
import json, os, sys
os.environ['DMC_USE_PROF'] = '1'
if os.environ.get('PLUGIN'):      # a profiling plugin (scripts/spec_variants.py with -DDMC_PROFILE=1) on the profiling library
  os.environ['DMC_NO_STATIC'] = '1'; os.environ['DMC_SPEC_PLUGIN'] = os.path.abspath(os.environ['PLUGIN'])
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
cfgid = int(os.environ.get('CONFIG', 5))
cfg = bench.CONFIGS[cfgid]
m = mc.compile_xml(common.read_model(cfg['asset'] + '.xml'))
B = int(os.environ.get('B', cfg['batch']))
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
print(b.info())
print('mean ncon', b.get('ncon').mean(), 'nefc mean', b.get('nefc').mean(), 'max', b.get('nefc').max(), 'iter', b.get('solver_iter').mean(), b.get('warning').sum(axis=0))
print('ms per launch (profiling build)', min(b.time_steps(nsub, 20) for _ in range(3)))
b.prof_enable(True)
N = 20
for t in range(N):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
p = b.prof_get()
tot = sum(p.values())
print('config %d: total cycles per physics step %.0f' % (cfgid, tot / N / nsub))
out = {}
for k, v in sorted(p.items(), key=lambda kv: -kv[1]):
  if v:
    print('   %-16s %9.0f  %5.1f%%' % (k, v / N / nsub, 100 * v / tot))
    out[k] = v / N / nsub
json.dump(dict(config=cfgid, cycles_per_physics_step=out, info=b.info()), open(os.path.join(ROOT, 'gpurun_out', 'phase_cfg%d.json' % cfgid), 'w'), indent=1)
