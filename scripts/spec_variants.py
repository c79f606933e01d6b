"""Builds the on-demand specialised kernel (dm_control_amd/specialise.py) of a BASELINE config's model once per flag set:
   python scripts/spec_variants.py 2 "" "-DDMC_NO_ROW_NEWBCAST"          NAME=<tag>: also copied to _spec_cache/v_<tag>_cfg<N>[_<k>].so
Run a variant with  DMC_NO_STATIC=1 DMC_SPEC_FLAGS="<flags>" python bench.py --config 2  (the baked kernel hidden, the plugin
of exactly these flags attached from the cache)."""
import os, shutil, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dm_control_amd import mjcf_compiler as mc, specialise
from dm_control_amd.suite import common
cfg = bench.CONFIGS[int(sys.argv[1])]
m = mc.compile_xml(common.read_model(cfg['asset'] + '.xml'))
caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {})); caps.pop('precision', None)
prec = int(os.environ.get('PRECISION', 32))
for k, flags in enumerate(sys.argv[2:] or ['']):
  os.environ['DMC_SPEC_FLAGS'] = flags
  t = time.time()
  p = specialise.warm(m, precision=prec, jlevel=(int(os.environ['JLEVEL']) if os.environ.get('JLEVEL') else None), **caps)      # JLEVEL=0: the kernel of a small batch (config 5)
  if os.environ.get('NAME'):
    q = os.path.join(os.path.dirname(p), 'v_%s_cfg%s%s.so' % (os.environ['NAME'], sys.argv[1], '_%d' % k if k else ''))
    shutil.copyfile(p, q); p = q
  print('%-40r %s  %.1f s' % (flags, os.path.basename(p), time.time() - t), flush=True)
