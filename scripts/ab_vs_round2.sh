# A/B against the round-2 library on ONE box: needs a worktree of the round-2 commit built in _r02/ (git worktree add _r02 45da38f; build there)
cd "$GRAFT_REPO_ROOT"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'value %.4g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)), d.get('workload_stats'), {k: (v.get('max'), v.get('frac_le_1e4')) for k, v in d.get('parity', {}).items() if isinstance(v, dict)})"; }
python -m pytest tests -q -m gpu 2>&1 | grep -E "^E  |passed|failed|FAILED" | head -20
for cfg in 2 4; do
  (cd _r02 && python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "r02 cfg$cfg")
  python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | show "now cfg$cfg"
done
