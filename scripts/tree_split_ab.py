"""Tree-split factorisations (StepDims::treemax) against the library built with -DDMC_NO_TREE_SPLIT: the end state of the
config-5 bench workload must be bit-identical (the split skips products with exact zeros only), and the rate of both."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
  import numpy as np
  import bench
  from dm_control_amd.batch import BatchedPhysics, OUT
  from dm_control_amd.suite import common
  cfg = bench.CONFIGS[int(os.environ.get('CONFIG', 5))]
  prec = int(os.environ.get('PREC', 32))
  m = bench.load_model(cfg['asset'])
  B = int(os.environ.get('B', cfg['batch']))
  caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {})); caps.pop('precision', None)
  b = BatchedPhysics(m, B, precision=prec, **caps)
  b.set('qpos', bench.initial_qpos(cfg, m, B, 0, phys=b))
  mask = 0
  for n in cfg['outputs']: mask |= OUT[n]
  b.set_output_mask(mask)
  b.forward(); b.sync()
  rs = np.random.RandomState(5)
  for t in range(int(os.environ.get('T', 200))):
    b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(cfg['nsub'])
  b.sync()
  ms = min(b.time_steps(cfg['nsub'], 50) for _ in range(3))
  np.savez(sys.argv[2], qpos=b.get('qpos'), qvel=b.get('qvel'), warm=b.get('qacc_warmstart'), ms=ms, it=b.get('solver_iter'))
  sys.exit(0)
import numpy as np
out = {}
for prec in (32, 64):
  res = {}
  for v in ('', 'nosplit'):
    f = os.path.join(ROOT, 'gpurun_out', 'split_%s_%d.npz' % (v or 'main', prec))
    env = dict(os.environ, DMC_LIB_VARIANT=v, PREC=str(prec))
    subprocess.check_call([sys.executable, __file__, 'child', f], env=env)
    res[v or 'main'] = np.load(f)
  a, b = res['main'], res['nosplit']
  out['f%d' % prec] = dict(bit_identical=bool(all(np.array_equal(a[k], b[k]) for k in ('qpos', 'qvel', 'warm'))),
                           max_abs_diff=float(max(np.max(np.abs(a[k] - b[k])) for k in ('qpos', 'qvel', 'warm'))),
                           ms_per_launch_split=float(a['ms']), ms_per_launch_full=float(b['ms']),
                           speedup=float(b['ms'] / a['ms']))
  print('f%d' % prec, out['f%d' % prec], flush=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'tree_split_ab.json'), 'w'), indent=1)
