#!/bin/bash
# A/B against the round-5 tree on ONE box: a worktree of the round-5 commit built in _r05/
# (git worktree add _r05 782cb49; python -c "import __graft_entry__ as g; g.build()" there).  CONFIGS="2 3 4 5" REPS=2
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'value %.5g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)), 'iter %.3f' % d['workload_stats']['mean_solver_iter'])"; }
{
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in ${CONFIGS:-2 3 4 5}; do
    (cd _r05 && DMC_BENCH_NO_PMC=1 python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 --pipeline 0 2>/dev/null | tail -1 | show "r05 cfg$cfg rep$rep")
    DMC_BENCH_NO_PMC=1 python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 --pipeline 0 --extra 0 2>/dev/null | tail -1 | show "r06 cfg$cfg rep$rep"
  done
done
} 2>&1 | tee gpurun_out/r06_ab_vs_round5.log
