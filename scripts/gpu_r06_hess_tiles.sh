#!/bin/bash
# Round 6: H = M + J'DJ of the 62-dof fp32 model assembled on the matrix cores and factored in place (hess_factor_tiles) --
# parity tests, then plugin twins on one box: tiles for assembly + factorisation / factorisation only / neither.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
export DMC_NO_STATIC=1 DMC_SPECIALISE=build
for f in "" "-DDMC_NO_HESS_TILES"; do
  DMC_SPEC_FLAGS="$f" timeout 900 python -m pytest tests/test_gpu_suite.py -m gpu -q -x -k "baseline_62dof" 2>&1 | tail -3
done
{
for rep in 1 2; do for c in ${CFGS:-4}; do
  for f in "" "-DDMC_NO_HESS_TILES" "-DDMC_NO_CHOL_TILES"; do
    DMC_SPEC_FLAGS="$f" DMC_BENCH_NO_PMC=1 timeout 400 python bench.py --config $c --no-cpu-baseline --parity-steps 20 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $c rep $rep', repr('$f'), 'value %.5g ms %.4f rollout %.5g iter %.3f' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter']), {k: '%.3g' % v['max'] for k, v in p.items() if isinstance(v, dict) and 'max' in v})"
  done
done; done
} 2>&1 | tee gpurun_out/r06_hess_tiles_ab.log
