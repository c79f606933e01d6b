cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for w in 5 6; do DMC_WAVES=$w DMC_BENCH_NO_PMC=1 timeout 400 python bench.py --config 4 --no-cpu-baseline --parity-steps 0 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); i=d['config']['info']
print('DMC_WAVES=$w value %.5g ms %.4f rollout %.5g envs_per_cu %s' % (d['value'], d['ms_per_step'], d['rollout']['value'], i['envs_per_cu']))"; done
