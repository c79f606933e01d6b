"""One gpurun call: GPU tests of the production library, then an A/B of instruction-scheduling variants of the same
sources (build.py variants: `-mllvm -amdgpu-sched-strategy=...`) on configs 2 / 5 / 4; if a variant wins config 2 by
more than 1.5 %, its GPU tests and the bench lines of configs 2-5 with it (-> gpurun_out/r03v_*).

  VARIANTS="maxilp" python scripts/gpu_r03_sched_ab.py
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
os.makedirs('gpurun_out', exist_ok=True)
T0 = time.time()
BUDGET = float(os.environ.get('BUDGET_S', '330'))
variants = ['-'] + os.environ.get('VARIANTS', 'maxilp').split()
log = {'runs': [], 'tests': {}}


def left():
  return BUDGET - (time.time() - T0)


def run(cmd, variant, timeout):
  env = dict(os.environ)
  env.pop('DMC_LIB_VARIANT', None)
  if variant != '-':
    env['DMC_LIB_VARIANT'] = variant
  try:
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
  except subprocess.TimeoutExpired:
    return None


def tests(variant):
  r = run([sys.executable, '-m', 'pytest', 'tests', '-m', 'gpu', '-x', '-q'], variant, 300)
  tail = (r.stdout.strip().splitlines() or ['timeout'])[-1] if r else 'timeout'
  log['tests'][variant] = tail
  print('tests', variant, ':', tail, '(%.0f s used)' % (time.time() - T0), flush=True)
  if r and r.returncode:
    print(r.stdout[-3000:], flush=True)
  return bool(r) and r.returncode == 0


def bench(config, variant, extra=()):
  r = run([sys.executable, 'bench.py', '--config', str(config)] + list(extra), variant, 200)
  if not r or r.returncode:
    print('bench failed', config, variant, (r.stderr[-800:] if r else 'timeout'), flush=True)
    return None
  return json.loads(r.stdout.strip().splitlines()[-1])


def save():
  with open('gpurun_out/r03_sched_ab.json', 'w') as f:
    json.dump(log, f, indent=1)


tests('-')
quick = ['--no-cpu-baseline', '--parity-steps', '0']
for config, reps in ((2, 3), (5, 2), (4, 1), (3, 1)):
  for rep in range(reps):
    for v in variants:
      d = bench(config, v, quick)
      if d:
        log['runs'].append(dict(config=config, variant=v, rep=rep, value=d['value'], ms=d['ms_per_step'],
                                rollout=d.get('rollout', {}).get('value')))
        print('config %d variant %-8s rep %d: %.4g env-steps/s, %.5f ms, rollout %.4g' % (
            config, v, rep, d['value'], d['ms_per_step'], d.get('rollout', {}).get('value') or 0), flush=True)
      save()


def mean(config, v):
  xs = [r['value'] for r in log['runs'] if r['config'] == config and r['variant'] == v]
  return sum(xs) / len(xs) if xs else 0.0


base = mean(2, '-')
best = max(variants, key=lambda v: mean(2, v))
log['summary'] = {str(c): {v: mean(c, v) for v in variants} for c in (2, 5, 4, 3)}
log['winner_config2'] = best
print('summary', json.dumps(log['summary']), 'winner on config 2:', best, '(%.0f s used)' % (time.time() - T0), flush=True)
save()
if best != '-' and base and mean(2, best) / base > 1.015 and left() > 60:
  ok = tests(best)
  save()
  if ok:
    for c in (2, 5, 4, 3):
      if left() < 50:
        print('out of budget before config', c, flush=True)
        break
      d = bench(c, best)
      if d:
        with open('gpurun_out/r03v_bench_cfg%d.json' % c, 'w') as f:
          json.dump(d, f)
        print('full bench config %d with %s: %.4g (%.0f s used)' % (c, best, d['value'], time.time() - T0), flush=True)
print('done in %.0f s' % (time.time() - T0))
