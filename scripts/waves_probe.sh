# Config 4 throughput against the environments resident per CU (DMC_WAVES forces the workgroup shape): the slope that a
# fifth resident environment would ride on.  Round 4, one box: 3 per CU 274.9 k, 4 per CU 328.5 k env-steps/s (+19.5 % for
# +33 % residency); two 2-wave workgroups instead of one 4-wave workgroup 318.2 k.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for w in 4 3 2 4; do
  DMC_WAVES=$w DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config 4 --no-cpu-baseline --parity-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves $w', 'value %.5g ms %.4f' % (d['value'], d['ms_per_step']), d['config']['info']['envs_per_cu'], d['config']['info']['grid'])"
done
