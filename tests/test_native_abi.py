"""The C-ABI shared library loads on a CPU-only box and exports every symbol
include/dmc_batch.h declares; compute entry points fail loudly without a GPU."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def native():
  from dm_control_amd import build
  build.build()
  from dm_control_amd import _native
  return _native


def test_header_symbols_are_exported(native):
  with open(os.path.join(ROOT, 'include', 'dmc_batch.h')) as f:
    hdr = f.read()
  declared = set(re.findall(r'\b(dmc_[a-z0-9_]+)\s*\(', hdr))
  assert declared, 'no declarations parsed'
  lib = native.lib()
  missing = [n for n in sorted(declared) if not hasattr(lib, n)]
  assert not missing, missing
  assert set(native.EXPORTS) == declared


def test_model_blob_validation(native):
  import ctypes
  import numpy as np
  lib = native.lib()
  bad_i = np.zeros(8, dtype=np.int32)
  bad_r = np.zeros(8)
  out = ctypes.c_void_p()
  rc = lib.dmc_model_create(bad_i.ctypes.data, 8, bad_r.ctypes.data, 8, ctypes.byref(out))
  assert rc != 0 and b'magic' in lib.dmc_last_error()


def test_no_cpu_fallback(native):
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from dm_control_amd import mjcf_compiler as mc
  from dm_control_amd.batch import BatchedPhysics
  with open(os.path.join(ROOT, 'dm_control_amd', 'suite', 'assets', 'cheetah.xml')) as f:
    m = mc.compile_xml(f.read())
  with pytest.raises(native.NativeError, match='no CPU fallback'):
    BatchedPhysics(m, 4)


def test_product_never_imports_oracle():
  # The oracle is test infrastructure: nothing under dm_control_amd/ may reference it.
  pkg = os.path.join(ROOT, 'dm_control_amd')
  for dirpath, _, files in os.walk(pkg):
    for fn in files:
      if fn.endswith(('.py', '.h', '.hip', '.cpp')):
        with open(os.path.join(dirpath, fn)) as f:
          txt = f.read()
        assert 'import oracle' not in txt and 'from oracle' not in txt, fn
        assert 'mjstep_oracle' not in txt.replace('oracle/mjstep_oracle.c)', ''), fn


def test_hot_kernel_keeps_its_locals_out_of_scratch(tmp_path):
  """Regression guard for profiles/r01_hbm_traffic.json: dynamically indexed locals / spills in
  the fused step kernel turn into HBM writes (17.8 MB per launch at one point).  The cheetah
  instantiation the bench runs may only use the few bytes its out-of-line calls need."""
  import subprocess
  from dm_control_amd import build
  build.build()
  # kernel metadata of the device code object inside the built (bundled) object file -- no recompilation
  blob = open(os.path.join(build.CSRC, 'step_kernels_f32.o'), 'rb').read()
  text, pos = '', blob.find(b'\x7fELF')
  while pos >= 0:
    elf = tmp_path / ('obj%d.elf' % pos)
    elf.write_bytes(blob[pos:])
    r = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', str(elf)], capture_output=True, text=True)
    if 'amdhsa.kernels' in r.stdout:
      text += r.stdout
    pos = blob.find(b'\x7fELF', pos + 4)
  sizes = dict(re.findall(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)', text))
  bench_kernel = [k for k in sizes if 'step_kernel_staticIfLi32ELi0ELb0' in k]      # (no claim loop: the batch fits the grid)
  assert bench_kernel, sorted(sizes)[:5]
  assert int(sizes[bench_kernel[0]]) <= 32, sizes[bench_kernel[0]]


def test_static_fp32_kernels_private_segment_stays_bounded():
  """Every model-specialised fp32 instantiation (configs 2-5 run these).  The small models' kernels spill no VGPR.  The
  large ones (static ids 5-8) keep the values that live across their out-of-line stage calls -- the kernel's I/O
  pointers -- in the private segment: stored once at kernel start, reloaded where an I/O stage needs them (the stage
  functions are `not_tail_called`, so LLVM's interprocedural register allocation lets them skip the callee-saved VGPR
  saves that used to cost 90 stores + 90 loads per call; `scripts/scratch_by_function.py` shows where every access is).
  What is left beside that: the dynamically indexed polygon / support arrays of the box-box / ellipsoid narrow phases in
  the position stage, touched only when such a pair is in range.  The helpers on the solver's inner loops
  (chol_factor_rows, chol_solve_rows, ls_eval_*) keep at most 20 bytes."""
  sys.path.insert(0, os.path.join(ROOT, 'scripts'))
  import kernel_resources
  from dm_control_amd import build
  build.build()
  ks = kernel_resources.kernels(os.path.join(build.CSRC, 'step_kernels_f32.o'))
  ilp = kernel_resources.kernels(os.path.join(build.CSRC, 'step_kernels_f32_ilp.o'))      # the large models' unit
  assert ilp and not set(ilp) & set(ks)
  # the two units must not share a host symbol with different bodies (launch_step_t<float> once was a weak symbol in
  # both: the linker kept the small-model unit's, and configs 4 / 5 ran the generic kernel)
  import subprocess
  def weak(obj):
    out = subprocess.run(['nm', '-C', os.path.join(build.CSRC, obj)], capture_output=True, text=True).stdout
    return {l.split(' ', 2)[2] for l in out.splitlines() if l[17:18] == 'W' and '__device_stub__' not in l}
  assert not weak('step_kernels_f32.o') & weak('step_kernels_f32_ilp.o')
  assert all(re.search(r'step_kernel_staticIfLi64ELi[5678]ELb', n) for n in ilp), sorted(ilp)
  ks.update(ilp)
  # static id -> bytes per lane (round 5, session 7: the branch-free row routines left 120 / 168 B on the 56- / 62-dof kernels, 168 / 200 B
  # with the work queue, from 212 / 324; id 8 = soccer 2v2 with everything in LDS, the layout of a batch of at most one environment per CU)
  # round 6: the box-box clipping polygons take 8 slots instead of 16 (ids 7 / 8: 690 -> 176 / 160 B, 240 / 224 B with the work queue);
  # the queued kernels copy their argument structs to LDS before the queue's first claim (ids 5 / 6 with the queue: 184 / 232 B; the
  # claim in front of the copies had parked both structs in scratch: 648 / 696 B)
  bound = {0: 32, 1: 32, 2: 640, 3: 32, 4: 32, 5: 208, 6: 240, 7: 256, 8: 256}
  spill = {5: 56, 6: 72, 7: 28, 8: 24}      # VGPRs the kernel body parks across the stage calls (none in the small models' kernels)
  seen = set()
  for name, r in ks.items():
    m = re.search(r'step_kernel_staticIfLi(\d+)ELi(\d+)ELb([01])', name)
    if not m:
      continue
    sid = int(m.group(2))
    seen.add(sid)
    assert r['vgpr_spill'] <= spill.get(sid, 0), (name, r)
    assert r['scratch'] <= bound[sid], (name, r)
    assert r['vgpr'] <= 256
  assert seen == set(bound)
