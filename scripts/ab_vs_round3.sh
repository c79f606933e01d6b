#!/bin/bash
# A/B against the round-3 library on ONE box: needs a worktree of the round-3 commit built in _r03/
# (git worktree add _r03 4b8cad2; python -c "import __graft_entry__ as g; g.build()" there).  CONFIGS="2 3 4 5" REPS=2
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'value %.5g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)), d.get('workload_stats'))"; }
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in ${CONFIGS:-2 3 4 5}; do
    (cd _r03 && python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "r03 cfg$cfg rep$rep")
    python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "now cfg$cfg rep$rep"
  done
done
