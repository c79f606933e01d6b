cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'value %.5g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)))"; }
for rep in 1 2 3; do
  for v in ${ORDER:-_r03 _lean .}; do
    (cd $v && DMC_BENCH_NO_PMC=1 python bench.py --config 2 --no-cpu-baseline --parity-steps 0 2>/dev/null | show "$v rep$rep")
  done
done
