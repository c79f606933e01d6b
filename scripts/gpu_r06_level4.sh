#!/bin/bash
# Round 6: offload level 4 (eleven arrays of the 62-dof models in the global scratch: a SIXTH resident environment per CU) --
# the 62-dof parity tests, then bench lines of configs 4 (and 3, 5, 2: unchanged layouts, six-wave launch bounds) on one box.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -n 4 -k "62dof or cmu or CMU or composer or go_to_target or baseline" 2>&1 | tail -4 | tee gpurun_out/r06_level4_tests.log
{
for rep in 1 2; do for c in ${CFGS:-4 3 5 2}; do
  DMC_BENCH_NO_PMC=1 timeout 400 python bench.py --config $c --no-cpu-baseline --parity-steps ${PSTEPS:-20} --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{}); i=d['config']['info']
print('cfg $c rep $rep value %.5g ms %.4f rollout %.5g iter %.3f envs_per_cu %s waves %s lds %s static %s' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter'], i['envs_per_cu'], i['waves_per_block'], i['lds_bytes_per_block'], i['static_id']), {k: '%.3g' % v['max'] for k, v in p.items() if isinstance(v, dict) and 'max' in v}, d['warnings_after_run'])"
done; done
} 2>&1 | tee gpurun_out/r06_level4_bench.log
