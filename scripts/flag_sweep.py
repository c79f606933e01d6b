"""Builds the library once per set of extra compiler flags (build.py variants) and A/Bs them against the production
library inside ONE gpurun call.  Round 3 found the instruction-scheduling strategy worth +7.5 % on config 4 (max-ilp for
the large models' unit, build.py); this is the harness for the rest of that search.

  build (CPU, ~5 min per variant):   python scripts/flag_sweep.py build
  measure (GPU box):                 python scripts/flag_sweep.py run  [CONFIGS="2 3" REPS=2]

Candidates (name -> flags); a variant that fails to build is skipped (iterative-ilp crashes clang 22 on the fp64 unit)."""
import json
import os
import subprocess
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CANDIDATES = {
    'memclause': ['-mllvm', '-amdgpu-sched-strategy=max-memory-clause'],
    'bias0': ['-mllvm', '-amdgpu-schedule-metric-bias=0'],
    'bias100': ['-mllvm', '-amdgpu-schedule-metric-bias=100'],
    'relaxocc': ['-mllvm', '-amdgpu-schedule-relaxed-occupancy=true'],
    'nopostra': ['-mllvm', '-enable-post-misched=false'],
    'maxilp': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],      # the whole library (round 3: config 2 -3.1 %, config 4 +7.5 %)
}


def build_all():
  from dm_control_amd import build
  ok = []
  for name, flags in CANDIDATES.items():
    try:
      build.build(variant=name, extra_flags=flags)
      ok.append(name)
    except Exception as e:  # pylint: disable=broad-except
      print('variant %s does not build: %s' % (name, str(e)[:200]), file=sys.stderr)
  print('built:', ' '.join(ok))


def run_all():
  os.chdir(ROOT)
  os.makedirs('gpurun_out', exist_ok=True)
  have = ['-'] + [n for n in CANDIDATES if os.path.exists(os.path.join(ROOT, 'dm_control_amd', 'libdmc_hip_%s.so' % n))]
  rows = []
  for config in os.environ.get('CONFIGS', '2').split():
    for rep in range(int(os.environ.get('REPS', 2))):
      for v in have:
        env = dict(os.environ)
        env.pop('DMC_LIB_VARIANT', None)
        if v != '-':
          env['DMC_LIB_VARIANT'] = v
        r = subprocess.run([sys.executable, 'bench.py', '--config', config, '--no-cpu-baseline', '--parity-steps', '0'],
                           env=env, capture_output=True, text=True)
        if r.returncode:
          print('bench failed:', config, v, r.stderr[-300:], flush=True)
          continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rows.append(dict(config=int(config), variant=v, rep=rep, value=d['value'], ms=d['ms_per_step']))
        print('config %s %-10s rep %d: %.4g env-steps/s, %.5f ms' % (config, v, rep, d['value'], d['ms_per_step']), flush=True)
        with open('gpurun_out/flag_sweep.json', 'w') as f:
          json.dump(rows, f, indent=1)


if __name__ == '__main__':
  (build_all if sys.argv[1:] == ['build'] else run_all)()
