#!/bin/bash
# Round 6: noslip's A in LDS for the small-batch layout (StepDims::nsalds) -- the library against its predecessor
# (dm_control_amd/libdmc_hip_nonsa.so, DMC_LIB_VARIANT=nonsa) on config 5, one box; then the soccer / noslip GPU tests.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2 3; do for v in "" nonsa; do
  DMC_LIB_VARIANT=$v DMC_BENCH_NO_PMC=1 timeout 400 python bench.py --config 5 --no-cpu-baseline --parity-steps 200 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg 5 rep $rep lib', repr('$v'), 'value %.5g ms %.4f rollout %.5g iter %.3f static %s lds %s' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter'], d['config']['info'].get('static_id'), d['config']['info'].get('lds_bytes_per_block')), {k: '%.3g' % v['max'] for k, v in p.items() if isinstance(v, dict) and 'max' in v})"
done; done
} 2>&1 | tee gpurun_out/r06_nsa_lds_ab.log
timeout 900 python -m pytest tests -m gpu -q -x -n 4 -k "soccer or noslip or composer or baseline" 2>&1 | tail -3 | tee -a gpurun_out/r06_nsa_lds_ab.log
