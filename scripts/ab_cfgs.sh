# A/B of two builds on ONE box over several BASELINE configs: ORDER = worktree directories (each with its own built
# library), CONFIGS = config numbers, REPS = repetitions (interleaved).  Prints one line per run.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'value %.5g ms %.5f' % (d['value'], d['ms_per_step']))"; }
for rep in $(seq 1 ${REPS:-2}); do
  for c in ${CONFIGS:-2 3 4 5}; do
    for v in ${ORDER:-_lean .}; do
      (cd $v && DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 2>/dev/null | show "cfg$c $v rep$rep")
    done
  done
done
