#!/bin/bash
# Round 6: the 62 x 62 fp32 Newton Hessian factored on the matrix cores (chol_factor_tiles) -- parity tests of the 62-dof /
# soccer models, then A/B on one box: the library's baked kernel (tiles) against plugin twins with and without the tiles.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_suite.py -m gpu -q -x -k "baseline_62dof or work_queue or sliced" 2>&1 | tail -5
{
for rep in 1 2; do for c in 4 5; do
  for f in "BAKED" "" "-DDMC_NO_CHOL_TILES"; do
    if [ "$f" = BAKED ]; then unset DMC_NO_STATIC DMC_SPEC_FLAGS; else export DMC_NO_STATIC=1 DMC_SPEC_FLAGS="$f"; fi
    DMC_BENCH_NO_PMC=1 timeout 400 python bench.py --config $c --no-cpu-baseline --parity-steps 20 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $c rep $rep', repr('$f'), 'value %.5g ms %.4f rollout %.5g iter %.3f' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter']), {k: '%.3g' % v['max'] for k, v in p.items() if isinstance(v, dict) and 'max' in v})"
  done
done; done
} 2>&1 | tee gpurun_out/r06_chol_tiles_ab.log
