#!/bin/bash
# round 5, session 2: GPU test suite on the new library; A/B against the round-4 library (dm_control_amd/libdmc_hip_r04.so,
# DMC_LIB_VARIANT=r04: same C-ABI) on ONE box; parity legs; generic kernel vs model-specialised; fp64 lines.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
show() { python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read()); print('$1', 'value %.5g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)), d.get('workload_stats'), d.get('parity', {}).get('summary', ''))
except Exception as e: print('$1 FAILED', e)"; }
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r05_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r05_gputests.log
for rep in 1 2; do
  for cfg in 2 3 4 5; do
    DMC_LIB_VARIANT=r04 python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "r04 cfg$cfg rep$rep"
    python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "new cfg$cfg rep$rep"
  done
done 2>&1 | tee gpurun_out/r05_ab_vs_round4.log
for cfg in 2 3 4 5; do
  DMC_BENCH_NO_PMC=1 python bench.py --config $cfg --no-cpu-baseline > gpurun_out/r05_parity_cfg$cfg.json 2> gpurun_out/r05_parity_cfg$cfg.err
  python -c "
import json; d=json.loads(open('gpurun_out/r05_parity_cfg$cfg.json').read()); print('parity cfg$cfg', json.dumps(d.get('parity'))[:1500])"
done 2>&1 | tee gpurun_out/r05_parity.log
for cfg in 2 3 4 5; do
  DMC_NO_STATIC=1 python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "generic cfg$cfg"
  python bench.py --config $cfg --precision 64 --no-cpu-baseline --parity-steps 0 2>/dev/null | tee gpurun_out/r05_bench_f64_cfg$cfg.json | show "fp64 cfg$cfg"
done 2>&1 | tee gpurun_out/r05_generic_and_fp64.log
