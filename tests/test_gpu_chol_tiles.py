"""chol_factor_tiles (step_core.h: the fp32 Newton Hessian of a 33 .. 64-dof model factored on the matrix cores) against an
fp64 factor of the same matrices and against chol_factor_rows, for matrix sizes on both sides of every tile boundary --
the production routines, compiled into scripts/chol_mfma_probe.hip.  The step kernels that use the routine are compared
with the oracle in test_gpu_suite.py (configs 4 / 5); this is the routine alone, including what it writes past the
packed triangle (nothing)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_tile_factorisation_matches_fp64_for_33_to_64_dofs(tmp_path):
  hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  exe = str(tmp_path / 'chol_probe')
  subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-result', '-o', exe,
                         os.path.join(ROOT, 'scripts', 'chol_mfma_probe.hip')])
  out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stdout + out.stderr
  rows = {}
  for line in out.stdout.splitlines():
    m = re.match(r'N (\d+), (\d+) wave\(s\) per CU, (\w+)\s*:\s*(\d+) cycles.*max error vs fp64 (\S+), words written past the triangle (\d+)', line)
    assert m, line
    rows[(int(m.group(1)), int(m.group(2)), m.group(3))] = (float(m.group(4)), float(m.group(5)), int(m.group(6)))
  print(out.stdout)
  assert sorted({k[0] for k in rows}) == [33, 40, 48, 49, 57, 62, 64]
  for (n, waves, which), (cycles, err, past) in rows.items():
    assert past == 0, (n, waves, which, past)
    assert err < 5e-6, (n, waves, which, err)      # measured: rows 1.1e-6 .. 1.2e-6, tiles 1.0e-6 .. 1.4e-6
  for waves in (1, 5):      # what the routine is for: the 62-dof walker (measured 37.1 k -> 15.3 k cycles alone on a CU)
    assert rows[(62, waves, 'tiles')][0] < 0.6 * rows[(62, waves, 'rows')][0]
