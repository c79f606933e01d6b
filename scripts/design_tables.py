"""Prints the measurement tables of DESIGN.md (sections 4 and 5) from profiles/r02_*.json."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: json.load(open(os.path.join(ROOT, 'profiles', n)))
names = {2: 'cheetah', 3: 'humanoid (5 substeps)', 4: 'CMU (6 substeps)', 5: 'soccer (5 substeps, B = 256)'}
print('| config | VALU / SALU / LDS / VMEM instr per wave per launch | wave time: issuing / parked on `s_waitcnt` | VALU issue slots used chip-wide | resident waves per CU | HBM-side bytes per launch (x algorithmic) |')
print('|---|---|---|---|---|---|')
for c in (2, 3, 4, 5):
  n, b = P('r02_pmc_cfg%d.json' % c), P('r02_bench_cfg%d.json' % c)
  w = n['SQ_WAVES']
  k = lambda x: ('%.1f k' % (x / 1e3)) if x < 1e6 else ('%.2f M' % (x / 1e6))
  info = b['config']['info']
  waves_cu = info['envs_per_cu'] * info['lanes_per_env'] // 64
  hb = n['hbm_bytes_per_launch']['total_corrected']
  print('| %d %s | %s / %s / %s / %s | %.0f %% / %.0f %% | %.0f %% | %d (%d envs)%s | %.1f MB (%.1fx) |' % (
      c, names[c], k(n['SQ_INSTS_VALU'] / w), k(n['SQ_INSTS_SALU'] / w), k(n['SQ_INSTS_LDS'] / w), k(n['SQ_INSTS_VMEM'] / w),
      100 * n['SQ_ACTIVE_INST_ANY'] / n['SQ_WAVE_CYCLES'], 100 * n['SQ_WAIT_ANY'] / n['SQ_WAVE_CYCLES'],
      100 * b['roofline_issue']['frac'], waves_cu, info['envs_per_cu'], ', work queue' if info.get('work_queue') else '',
      hb / 1e6, hb / b['roofline']['algorithmic_bytes_per_launch']))
print()
print('| config | env-steps/s (physics steps/s) | ms / launch | rollout mode | CPU baseline (256 cores) | fp32 teacher-forced median / max | fp32 open loop max | fp64 kernel open loop max | warnings |')
print('|---|---|---|---|---|---|---|---|---|')
wl = {2: 'cheetah run, B = 4096', 3: 'humanoid stand, B = 4096, 5 substeps', 4: 'CMU humanoid on Floor, B = 4096, 6 substeps', 5: 'soccer 2v2 BoxHead, B = 256, 5 substeps'}
for c in (2, 3, 4, 5):
  b = P('r02_bench_cfg%d.json' % c)
  f = lambda x: ('%.2f M' % (x / 1e6)) if x >= 1e6 else ('%.1f k' % (x / 1e3))
  pa = b['parity']
  print('| %d %s | **%s** (%s) | %.3g | %s | %s | %.1e / %.1e | %.1e (%d steps) | %.1e | %d |' % (
      c, wl[c], f(b['value']), f(b['physics_steps_per_s']), b['ms_per_step'], f(b['rollout']['value']), f(b['cpu_baseline']['value']),
      pa['teacher-forced']['median'], pa['teacher-forced']['max'], pa['open-loop']['max'], pa['steps'] * pa['n_sub_steps'],
      pa['f64-open-loop']['max'], sum(b['warnings_after_run'])))
print()
for n in ('cheetah', 'humanoid', 'cmu_2019_position_floor', 'soccer_2v2_boxhead'):
  d = P('r02_phase_%s.json' % n)['cycles_per_physics_step']
  tot = sum(d.values())
  print(n, '%.0f k cycles per physics step per wave:' % (tot / 1e3), ', '.join('%s %.0f %%' % (a, 100 * v / tot) for a, v in sorted(d.items(), key=lambda kv: -kv[1])[:9]))
print()
for r in P('r02_composer_runs.json'):
  print(r['env'], r['B'], r['mode'], 'fused' if r.get('fused') else 'per-substep launches', '%.1f k env-steps/s' % (r['env_steps_per_s'] / 1e3), 'warnings', sum(r['warnings']))
