# Config 4 throughput against the environments resident per CU (DMC_WAVES forces the workgroup shape).
# Round 4, one box, BEFORE the level-3 offload: 3 per CU 274.9 k, 4 per CU 328.5 k env-steps/s (+19.5 % for +33 %
# residency); two 2-wave workgroups instead of one 4-wave workgroup 318.2 k.  With the level-3 library the same probe
# separates what the fifth environment brings from what the offload costs (profiles/r04_waves_probe_level3.log).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for w in ${WAVES:-5 4 3 5 4}; do
  DMC_WAVES=$w DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config 4 --no-cpu-baseline --parity-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves $w', 'value %.5g ms %.4f' % (d['value'], d['ms_per_step']), 'envs/CU', d['config']['info']['envs_per_cu'], 'grid', d['config']['info']['grid'])"
done
