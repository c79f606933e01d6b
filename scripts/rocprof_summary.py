"""Summarises a rocprofv3 results .db (kernel trace) as text: per kernel calls,
min/median/avg/max duration, registers, LDS, scratch, grid."""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, end-start, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels").fetchall()
byname = {}
for r in rows:
  byname.setdefault(r[0], []).append(r)
tot = sum(r[1] for r in rows)
print('%-60s %6s %10s %10s %10s %10s %6s | vgpr agpr sgpr lds scratch grid wg' % ('kernel', 'calls', 'min_us', 'med_us', 'avg_us', 'max_us', 'pct'))
for name, rs in sorted(byname.items(), key=lambda kv: -sum(r[1] for r in kv[1])):
  d = np.array([r[1] for r in rs]) / 1e3
  r0 = rs[0]
  short = name if len(name) < 60 else name[:57] + '...'
  print('%-60s %6d %10.1f %10.1f %10.1f %10.1f %6.2f | %d %d %d %d %d %d %d' % (
      short, len(rs), d.min(), np.median(d), d.mean(), d.max(), 100 * d.sum() * 1e3 / tot,
      r0[2], r0[3], r0[4], r0[5], r0[6], r0[7], r0[8]))
try:
  pm = cur.execute("select * from pmc_events limit 1").fetchall()
  if pm:
    cols = [d[1] for d in cur.execute("pragma table_info(pmc_events)")]
    print('pmc columns:', cols)
except Exception as e:  # pylint: disable=broad-except
  print('no pmc:', e)
