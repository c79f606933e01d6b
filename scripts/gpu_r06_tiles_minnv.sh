#!/bin/bash
# Round 6: chol_factor_tiles for the 17 .. 32-dof fp32 models on one wave per environment (humanoid 27, soccer 30 when its
# Hessian is not split)?  Plugin twins on one box: DMC_TILES_MIN_NV = 33 (default) against 17.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
export DMC_NO_STATIC=1 DMC_SPECIALISE=build
{
for rep in 1 2; do for c in ${CFGS:-3 5 4}; do
  for f in "" "-DDMC_TILES_MIN_NV=17"; do
    [ $c = 4 ] && [ -n "$f" ] && continue
    DMC_SPEC_FLAGS="$f" DMC_BENCH_NO_PMC=1 timeout 400 python bench.py --config $c --no-cpu-baseline --parity-steps 20 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $c rep $rep', repr('$f'), 'value %.5g ms %.4f rollout %.5g iter %.3f static %s' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter'], d.get('launch',{}).get('static_id')), {k: '%.3g' % v['max'] for k, v in p.items() if isinstance(v, dict) and 'max' in v})"
  done
done; done
} 2>&1 | tee gpurun_out/r06_tiles_minnv_ab.log
