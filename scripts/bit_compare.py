"""Are two builds of the library bit-identical on a model?  MODEL=<asset> VARIANTS="main base" PRECISIONS="32 64": each
variant (DMC_LIB_VARIANT) steps the same seeded batch in its own process; the final (qpos, qvel, sensordata) are compared."""
import hashlib, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
  import bench
  from dm_control_amd import mjcf_compiler as mc
  from dm_control_amd.batch import BatchedPhysics, OUT
  from dm_control_amd.suite import common
  cfg = bench.CONFIGS[int(os.environ['CONFIG'])]
  m = mc.compile_xml(common.read_model(cfg['asset'] + '.xml'))
  B = 64
  caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {})); caps.pop('precision', None)
  b = BatchedPhysics(m, B, precision=int(os.environ['PRECISION']), **caps)
  b.set('qpos', bench.initial_qpos(cfg, m, B, 0, phys=b))
  mask = 0
  for n in cfg['outputs']: mask |= OUT[n]
  b.set_output_mask(mask)
  rs = np.random.RandomState(7)
  for t in range(int(os.environ.get('STEPS', 100))):
    b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(cfg['nsub'])
  h = hashlib.sha256()
  for f in ('qpos', 'qvel', 'sensordata'):
    h.update(np.ascontiguousarray(b.get(f)).tobytes())
  print('HASH', h.hexdigest(), float(np.abs(b.get('qpos')).sum()))
  sys.exit(0)
for c in os.environ.get('CFGS', '2').split():
  for prec in os.environ.get('PRECISIONS', '32 64').split():
    res = {}
    for v in os.environ.get('VARIANTS', 'main base').split():
      env = dict(os.environ, CONFIG=c, PRECISION=prec, DMC_LIB_VARIANT='' if v == 'main' else v)
      out = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
      line = [l for l in out.stdout.splitlines() if l.startswith('HASH')]
      res[v] = line[0] if line else 'FAILED ' + out.stderr[-300:]
    vals = list(res.values())
    print('config', c, 'fp%s' % prec, 'IDENTICAL' if all(x == vals[0] for x in vals) else 'DIFFERENT', res)
