"""Where the wave-time of a queued launch goes (CONFIG = 3 / 4): from the wave trace of one launch per env-step under the
bench workload -- start / end clock of every item, the workgroup it ran on -- the number of items in flight over time, the
utilisation of the resident wave slots, and how well the longest-first hand-out (previous launch's cost) predicts this
launch's durations."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
cfgid = int(os.environ.get('CONFIG', 4))
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
info = b.info()
resident = info['grid'] * info['waves_per_block']
res = dict(config=cfgid, info=info, resident_waves=resident, ms_launch=min(b.time_steps(nsub, 20) for _ in range(3)))
b.wave_trace(True)
out = []
prev = None
for rep in range(4):
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  for _ in range(8):
    b.step(nsub)
  b.sync()
  tr = b.wave_trace().astype(np.int64)
  k = 7
  start, end, wg = tr[k, 1], tr[k, 2], tr[k, 3]
  t0 = tr[k, 0].min()
  dur = end - start
  span = int(end.max() - t0)
  ev = np.concatenate([np.stack([start - t0, np.ones_like(start)], 1), np.stack([end - t0, -np.ones_like(end)], 1)])
  ev = ev[np.argsort(ev[:, 0], kind='stable')]
  active = np.cumsum(ev[:, 1])
  t = ev[:, 0]
  def first_below(frac):
    idx = np.nonzero((active < frac * resident) & (t > 0.3 * span))[0]
    return float(t[idx[0]]) / span if idx.size else 1.0
  # time-weighted mean number of items in flight
  mean_active = float(np.sum(active[:-1] * np.diff(t)) / span)
  order = np.argsort(start, kind='stable')      # hand-out order
  late = start - t0 > 0.6 * span
  prev_dur = tr[k - 1, 2] - tr[k - 1, 1]
  rec = dict(span_ticks=span, work_ticks=int(dur.sum()), utilisation=float(dur.sum() / (resident * span)), mean_items_in_flight=mean_active,
             ideal_span_over_span=float(dur.sum() / resident / span), max_item_over_span=float(dur.max() / span),
             t_below_90pct=first_below(0.9), t_below_50pct=first_below(0.5), t_below_10pct=first_below(0.1),
             first_wave_start=float((start.min() - t0) / span), last_start=float((start.max() - t0) / span),
             corr_prev_cost_vs_dur=float(np.corrcoef(prev_dur, dur)[0, 1]),
             late_items=int(late.sum()), late_items_longer_than_median=int((late & (dur > np.median(dur))).sum()),
             dur_of_last_20_started=[int(x) for x in dur[order[-20:]]],
             items_per_workgroup_minmax=[int(np.bincount(wg).min()), int(np.bincount(wg).max())])
  out.append(rec)
  print(json.dumps(rec))
res['launches'] = out
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'occupancy_cfg%d.json' % cfgid), 'w'), indent=1)
