"""Prints the LDS footprint (tables + per-environment scratch, array by array) of the BASELINE models.

Host only (no GPU): runs the MJCF compiler and dm_control_amd/csrc/lds_report.cpp.
  python scripts/lds_report.py [model[:nconmax[:njmax]] ...]
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler  # noqa: E402

CSRC = os.path.join(ROOT, 'dm_control_amd', 'csrc')
DEFAULT = ['cheetah', 'humanoid:24', 'humanoid_CMU:32', 'cmu_2019_position_floor:32', 'soccer_2v2_boxhead:24']


def report(spec):
  parts = spec.split(':')
  name = parts[0]
  caps = [int(x) for x in parts[1:]] + [0, 0, 0]
  tool = os.path.join(CSRC, 'lds_report')
  src = tool + '.cpp'
  deps = [src, os.path.join(CSRC, 'step_tables.h'), os.path.join(CSRC, 'step_layout.h')]
  if not os.path.exists(tool) or os.path.getmtime(tool) < max(os.path.getmtime(d) for d in deps):
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-o', tool, src])
  with open(os.path.join(ROOT, 'dm_control_amd', 'suite', 'assets', name + '.xml')) as f:
    m = mjcf_compiler.compile_xml(f.read())
  ints, reals = m.pack()
  with tempfile.TemporaryDirectory() as td:
    fi, fr = os.path.join(td, 'i.bin'), os.path.join(td, 'r.bin')
    ints.tofile(fi)
    reals.tofile(fr)
    out = subprocess.check_output([tool, name, fi, fr, str(caps[0]), str(caps[1]), str(caps[2])]).decode()
  return json.loads(out)


def main():
  specs = sys.argv[1:] or DEFAULT
  for spec in specs:
    r = report(spec)
    hdr = {k: v for k, v in r.items() if not isinstance(v, dict)}
    print(json.dumps(hdr))
    for sect in ('int_tables', 'real_tables', 'scratch_real', 'ovl_pos', 'ovl_vel', 'ovl_sol', 'scratch_int'):
      items = sorted(r[sect].items(), key=lambda kv: -kv[1])
      tot = sum(v for _, v in items)
      print('  %-12s %6d words (%5.1f KiB fp32): %s' % (sect, tot, tot*4/1024.0, ', '.join('%s=%d' % kv for kv in items[:10])))
    tables = (r['n_mi'] + r['n_mr_lds'])*4
    env = (r['n_sr'] + r['n_si'])*4
    print('  fp32: tables %.1f KiB, env %.1f KiB -> %d envs per 160 KiB CU; global scratch %.1f KiB per env' % (tables/1024.0, env/1024.0, (160*1024 - tables)//env, r['n_gs']*4/1024.0))


if __name__ == '__main__':
  main()
