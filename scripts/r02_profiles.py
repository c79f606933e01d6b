"""Summarises gpurun_out/{prof,pmcA,pmcB,pmcF,pmcW}_cfg<N> (rocprofv3 csv) into profiles/r02_*: per launch kind
(single env-step launches / rollout launches / settle) kernel durations, SQ counters and HBM bytes per launch."""
import collections, csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
STEPS = {2: 300, 3: 30, 4: 30, 5: 30}


def rows(pattern):
  for f in glob.glob(os.path.join(ROOT, 'gpurun_out', pattern), recursive=True):
    with open(f) as fh:
      for r in csv.DictReader(fh):
        yield r


lines = []
for c in (2, 3, 4, 5):
  K = STEPS[c]
  trace = [r for r in rows('prof_cfg%d/**/*kernel_trace.csv' % c) if 'step_kernel' in r['Kernel_Name']]
  trace.sort(key=lambda r: int(r['Start_Timestamp']))
  dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in trace]
  if not dur:
    continue
  # launch order of bench.py: init / settle launches, 5 warm-up steps, K timed single-step launches, then rollout launches
  name = trace[0]['Kernel_Name'].split('(')[0][-60:]
  med = statistics.median(dur)
  single = [d for d in dur if d < 3 * med]          # single env-step launches (settle / rollout launches are 10-300 x longer)
  other = [d for d in dur if d >= 3 * med]
  out = dict(config=c, kernel=name, launches=len(dur), single_step=dict(n=len(single), min_us=min(single), median_us=statistics.median(single),
             avg_us=sum(single) / len(single), max_us=max(single)),
             multi_step=dict(n=len(other), median_us=statistics.median(other) if other else None, max_us=max(other) if other else None))
  pmc = {}
  for tag in ('A', 'B', 'F', 'W'):
    acc = collections.defaultdict(list)
    tr = {}
    for r in rows('pmc%s_cfg%d/**/*kernel_trace.csv' % (tag, c)):
      tr[r.get('Dispatch_Id')] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for r in rows('pmc%s_cfg%d/**/*counter_collection.csv' % (tag, c)):
      if 'step_kernel' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append((float(r['Counter_Value']), tr.get(r.get('Dispatch_Id'))))
    for k, v in acc.items():
      vals = [x for x, _ in v]
      m = statistics.median(vals)
      sel = [x for x in vals if x < 3 * m] or vals          # single-step launches only
      pmc[k] = statistics.median(sel)
      d = [t for x, t in v if t is not None and x < 3 * m]
      if d:
        pmc['kernel_us_pass' + tag] = statistics.median(d)
  if pmc:
    out['pmc_single_step_launch'] = pmc
    prof = dict(pmc)
    prof['kernel_us'] = pmc.get('kernel_us_passA', out['single_step']['median_us'])
    if 'GRBM_GUI_ACTIVE' in pmc and prof['kernel_us']:
      prof['clock_ghz'] = pmc['GRBM_GUI_ACTIVE'] / 8 / (prof['kernel_us'] * 1e3)      # GRBM counter is summed over the 8 XCDs
    if 'FETCH_SIZE' in pmc and 'WRITE_SIZE' in pmc:
      # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM): x2 on reads
      prof['hbm_bytes_per_launch'] = dict(fetch_raw=pmc['FETCH_SIZE'] * 1024, fetch_corrected=2 * pmc['FETCH_SIZE'] * 1024,
                                          write=pmc['WRITE_SIZE'] * 1024, total_corrected=(2 * pmc['FETCH_SIZE'] + pmc['WRITE_SIZE']) * 1024)
    json.dump(prof, open(os.path.join(ROOT, 'profiles', 'r02_pmc_cfg%d.json' % c), 'w'), indent=1)
  json.dump(out, open(os.path.join(ROOT, 'profiles', 'r02_kernel_stats_cfg%d.json' % c), 'w'), indent=1)
  lines.append(json.dumps(out))
  try:
    b = json.load(open(os.path.join(ROOT, 'gpurun_out', 'bench_cfg%d.json' % c)))
    json.dump(b, open(os.path.join(ROOT, 'profiles', 'r02_bench_cfg%d.json' % c), 'w'), indent=1)
  except Exception as ex:  # pylint: disable=broad-except
    print('no bench json for cfg', c, ex)
print('\n'.join(lines))
