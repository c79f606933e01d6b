"""The cross-lane layer of step_core.h on the device, in isolation (scripts/lane_primitives_probe.hip builds the production
routines): group_sum / group_max / group_scan / wave_bcast / the row_newbcast form of bcast_rows for 16, 32 and 64 lanes per
environment against host sums, maxima, prefix sums and lane picks (exact: the inputs make every partial sum exact); the
row-per-lane Cholesky factorisation and substitution (v_readlane / DPP broadcasts, several environments per wave) against an
fp64 solve and, bit for bit, against the fenced LDS forms they replaced; the per-tree routines against the whole-matrix ones
on a block-diagonal matrix, bit for bit.  The CPU tier cannot reach any of this (tests/emu runs one lane per environment)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cross_lane_primitives_and_row_linear_algebra(tmp_path):
  hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  exe = str(tmp_path / 'lane_probe')
  subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-result', '-o', exe,
                         os.path.join(ROOT, 'scripts', 'lane_primitives_probe.hip')], stderr=subprocess.DEVNULL)
  out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
  print(out.stdout)
  rows = []
  for line in out.stdout.splitlines():
    if line.startswith('failures'):
      continue
    m = re.match(r'(\w+) lpe (\d+) n (\d+) (\S+): max_err (\S+) mismatches (\d+)', line)
    assert m, line
    rows.append((m.group(1), int(m.group(2)), int(m.group(3)), m.group(4), float(m.group(5)), int(m.group(6))))
  names = {r[0] for r in rows}
  assert names == {'group_sum', 'group_max', 'group_scan', 'wave_bcast', 'bcast_rows16', 'chol_rows_vs_fp64_and_lds', 'chol_trees_vs_rows'}
  assert {r[1] for r in rows if r[0] == 'group_sum'} == {16, 32, 64}
  for name, lpe, n, typ, err, mism in rows:
    if name.startswith('chol_rows'):
      assert err < (2e-4 if typ == 'f32' else 1e-11), (name, lpe, n, typ, err)
      # fp64 keeps the order of the fenced LDS routines operation for operation; fp32 contracts a - l * l into an fma where
      # the row form allows it (nmsub<N <= 16>), so its factor may differ from the LDS form's by rounding
      if typ == 'f64':
        assert mism == 0, (name, lpe, n, typ, mism)
    else:
      assert mism == 0, (name, lpe, n, typ, mism)
      if name.startswith('chol_trees'):
        assert err < (2e-4 if typ == 'f32' else 1e-11)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
