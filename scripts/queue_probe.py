"""Work-queue utilisation of a config-3 / config-4 launch (CONFIG=3|4): per workgroup, when its last item ends; per
launch, the busy fraction of the resident wave slots (sum of item durations / (resident waves x span)) and how good the
longest-first order was (rank correlation of the previous launch's item cost with this launch's item duration).
Writes gpurun_out/queue_probe_cfg<N>.json.

The probe looks at WHOLE items (DMC_SLICES=1 unless set): with sliced items (the default since round 6) an item's trace
row spans its first piece's start to its last piece's end, waits included, and says nothing about wave occupancy.  Its
`makespan_substep_pieces_round_robin` is the prediction the sliced queue was built on (config 4: 903 k ticks -> 691 k
predicted; measured after the change: 8.90 -> 7.09 ms per launch)."""
import json, os, sys
os.environ.setdefault('DMC_SLICES', '1')
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
for t in range(int(os.environ.get('WARM', 100))):
  b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
b.sync()
info = b.info()
res = dict(config=cfgid, info=info)
res['ms_launch'] = min(b.time_steps(nsub, 20) for _ in range(3))
b.wave_trace(True)
wpb = info['waves_per_block']
nres = info['grid'] * wpb
out = []
prev = None
for rep in range(4):
  b.set_control(rs.uniform(-1, 1, (B, m.nu)))
  for _ in range(8):
    b.step(nsub)
  b.sync()
  tr = b.wave_trace().astype(np.int64)
  k = 7
  ent0 = tr[k, 0].min()
  start, end, blk = tr[k, 1] - ent0, tr[k, 2] - ent0, tr[k, 3]
  dur = end - start
  span = int(end.max())
  # end of the last item of every workgroup (a lower bound of when its last wave went idle)
  last = np.array([end[blk == g].max() for g in range(info['grid'])])
  # when does the queue run dry: the latest START of any item
  d = dict(span=span, busy_frac=float(dur.sum() / (nres * span)), sum_item_ticks=int(dur.sum()), resident_waves=int(nres),
           ideal_span=float(dur.sum() / nres), longest_item=int(dur.max()), median_item=float(np.median(dur)),
           last_start=int(start.max()), first_round_start_max=int(np.sort(start)[nres - 1]),
           wg_last_end_pct=np.percentile(last, [0, 10, 50, 90, 100]).tolist(),
           items_started_after_80pct=int((start > 0.8 * span).sum()),
           dur_of_items_ending_last=[int(x) for x in dur[np.argsort(end)[-10:]]],
           start_of_items_ending_last=[int(x) for x in start[np.argsort(end)[-10:]]])
  # launch k - 1 of the ring is the CONSECUTIVE predecessor: its item durations are what the longest-first order of
  # launch k was built from
  pdur = tr[k - 1, 2] - tr[k - 1, 1]
  rk = lambda x: np.argsort(np.argsort(x))
  d['rank_corr_dur_with_consecutive_predecessor'] = float(np.corrcoef(rk(pdur), rk(dur))[0, 1])
  d['rank_corr_start_with_predecessor_dur'] = float(np.corrcoef(rk(pdur), rk(start))[0, 1])      # -1 = handed out longest first
  d['mean_abs_rel_change_consecutive'] = float(np.mean(np.abs(dur - pdur) / pdur))
  # what a perfect longest-first order of THIS launch's durations would give (greedy list scheduling on nres slots)
  import heapq
  for name, seq in (('lpt_oracle', np.argsort(-dur)), ('lpt_predecessor', np.argsort(-pdur)), ('index_order', np.arange(dur.size))):
    h = [0] * nres
    heapq.heapify(h)
    for i in seq:
      heapq.heappush(h, heapq.heappop(h) + int(dur[i]))
    d['makespan_' + name] = int(max(h))
  # the same with items cut at substep boundaries, one substep = dur / nsub, handed out round by round (substep s of
  # every item before substep s + 1 of any), a piece starting no earlier than its predecessor's end
  piece = dur / nsub
  h = [(0, w) for w in range(nres)]
  heapq.heapify(h)
  ready = np.zeros(dur.size)
  for s_ in range(nsub):
    for i in np.argsort(-pdur):
      t, w = heapq.heappop(h)
      t = max(t, ready[i]) + piece[i]
      ready[i] = t
      heapq.heappush(h, (t, w))
  d['makespan_substep_pieces_round_robin'] = int(max(t for t, _ in h))
  out.append(d)
res['launches'] = out
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'queue_probe_cfg%d.json' % cfgid), 'w'), indent=1)
