"""Copies the summaries of the round-6 closing session (scripts/gpu_r06_final.sh) from gpurun_out/ into profiles/:
bench lines, rocprofv3 kernel stats (the csv and a per-launch-kind json: single env-step launches / rollout / settle),
device environments, soak, wave tails, the host-buffer boundary, logs."""
import csv, glob, json, os, shutil, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
PRE = os.environ.get('OUT_PREFIX', 'r06')      # a second session of the round is filed under another prefix (r06b)
OUT = lambda f: os.path.join(P, f.replace('r06_', PRE + '_', 1))
for f in ['r06_bench_cfg%d.json' % c for c in (2, 3, 4, 5)] + ['r06_bench_driver_cmd.json', 'r06_composer_runs.json', 'r06_soak_all_tasks.json',
         'r06_pcie_probe_cfg2.json', 'r06_gputests.log', 'r06_reference_on_hip.log', 'r06_config_runs.json', 'r06_smoke.log', 'r06_fused_env_runs.json',
         'r06_queue_probe_cfg3.json', 'r06_queue_probe_cfg4.json', 'r06_bench_driver_cmd_pmc.json', 'r06_ab_vs_round5.log', 'r06_2rank_gloo_single_device.json', 'r06_fused_env_runs.log'] + ['r06_bench_f64_cfg%d.json' % c for c in (2, 3, 4, 5)] + ['r06_wave_tail_cfg%d.json' % c for c in (2, 3, 4, 5)]:
  if os.path.exists(os.path.join(G, f)):
    shutil.copy(os.path.join(G, f), OUT(f))
for c in (2, 3, 4, 5):
  stats = glob.glob(os.path.join(G, 'r06_prof_cfg%d' % c, '**', '*kernel_stats.csv'), recursive=True)
  if stats:
    shutil.copy(stats[0], OUT('r06_rocprof_kernel_stats_cfg%d.csv' % c))
  rows = []
  for f in glob.glob(os.path.join(G, 'r06_prof_cfg%d' % c, '**', '*kernel_trace.csv'), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if 'step_kernel' in r['Kernel_Name']]
  if not rows:
    continue
  rows.sort(key=lambda r: int(r['Start_Timestamp']))
  dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
  try:
    line = json.load(open(os.path.join(G, 'r06_prof_bench_cfg%d.json' % c)))
  except Exception:  # pylint: disable=broad-except
    line = {}
  K, W = line.get('steps'), line.get('warmup')
  med = statistics.median(dur)
  single = [d for d in dur if d < 3 * med]
  other = [d for d in dur if d >= 3 * med]
  out = dict(config=c, kernel=rows[0]['Kernel_Name'].split('(')[0][-70:], launches=len(dur), bench_line=dict(steps=K, warmup=W, ms_per_step=line.get('ms_per_step'),
             kernel_ms_avg_hip_events=(line.get('roofline') or {}).get('kernel_ms_avg')),
             single_step_launches=dict(n=len(single), min_us=min(single), median_us=statistics.median(single), avg_us=sum(single) / len(single), max_us=max(single)),
             longer_launches=dict(n=len(other), median_us=statistics.median(other) if other else None, max_us=max(other) if other else None,
                                  note='the settle launch(es) and the rollout-mode launches of the same kernel'))
  if K:
    # the K timed launches are the K single-step launches right before the FIRST rollout-mode launch (what follows the
    # rollout launches is the pipelined leg: part-batches on two streams, not the timed whole-batch launches)
    first_long = next((i for i, d in enumerate(dur) if d >= 3 * med and i >= K), len(dur))
    idx = [i for i in range(first_long) if dur[i] < 3 * med]
    timed = [dur[i] for i in idx[-K:]] if len(idx) >= K else single
    out['timed_launches'] = dict(n=len(timed), avg_us=sum(timed) / len(timed), median_us=statistics.median(timed), max_us=max(timed))
  json.dump(out, open(OUT('r06_kernel_stats_cfg%d.json' % c), 'w'), indent=1)
  print('cfg', c, out.get('timed_launches'), out['bench_line'])
