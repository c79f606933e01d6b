"""Builds libdmc_hip.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  The library is git-ignored but travels to
the GPU box with the working-tree snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdmc_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'

_COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
_UNITS = [
    ('step_kernels_f32.hip', []),                       # default fp contraction (fma)
    ('step_kernels_f64.hip', ['-ffp-contract=off']),    # rounds like the fp64 oracle
    ('dmc_api.hip', []),
]
_DEPS = ['step_core.h', 'step_layout.h', 'step_tables.h', 'step_kernel.hip.h',
         '../../include/dmc_model_layout.h', '../../include/dmc_batch.h']


def _stale(target, sources):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False, profile=False, variant=None, extra_flags=()):
  """profile=True builds libdmc_hip_prof.so with per-phase cycle counters
  (-DDMC_PROFILE) next to the production library.  variant/extra_flags build an
  experimental libdmc_hip_<variant>.so (tuning studies; select it at run time
  with DMC_LIB_VARIANT=<variant>)."""
  deps = [os.path.join(CSRC, d) for d in _DEPS]
  objs = []
  procs = []
  tag = 'prof' if profile else variant
  lib = LIB.replace('.so', '_%s.so' % tag) if tag else LIB
  for src, flags in _UNITS:
    s = os.path.join(CSRC, src)
    o = os.path.join(CSRC, src.replace('.hip', '_%s.o' % tag if tag else '.o'))
    flags = flags + list(extra_flags)
    if profile:
      flags = flags + ['-DDMC_PROFILE=1']
    objs.append(o)
    if force or _stale(o, [s] + deps):
      cmd = [HIPCC] + _COMMON + flags + ['-c', s, '-o', o]
      if verbose:
        print(' '.join(cmd), file=sys.stderr)
      procs.append((cmd, subprocess.Popen(cmd)))
  for cmd, p in procs:
    if p.wait() != 0:
      raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
  if force or procs or _stale(lib, objs):
    cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', lib] + objs
    if verbose:
      print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
  return lib


if __name__ == '__main__':
  var = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--variant=')]
  xf = [a for a in sys.argv if a.startswith('-D') or a.startswith('-m')]
  print(build(force='--force' in sys.argv, verbose=True, profile='--profile' in sys.argv,
              variant=var[0] if var else None, extra_flags=xf))
