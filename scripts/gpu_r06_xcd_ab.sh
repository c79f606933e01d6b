#!/bin/bash
# Round 6: first round of a sliced launch as ONE queue whose claims bind items to XCDs, against the static deal
# (libdmc_hip_xstat.so = the library of commit cf913da), one box.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_suite.py -m gpu -q -x -k "work_queue or sliced" 2>&1 | tail -3
{
for rep in 1 2 3; do for c in 4 3; do for v in xstat ""; do
  if [ -n "$v" ]; then export DMC_LIB_VARIANT=$v; else unset DMC_LIB_VARIANT; fi
  DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg $c %-12s value %.5g ms %.4f rollout %.5g' % ('static deal' if '$v' else 'bound at claim', d['value'], d['ms_per_step'], d['rollout']['value']))"
done; done; done
unset DMC_LIB_VARIANT
for c in 4 3; do CONFIG=$c timeout 300 python scripts/xcd_balance_probe.py | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg $c bound at claim:', [(x['items_finished_per_xcd'], round(x['mean_over_max'],3)) for x in d])"; done
} 2>&1 | tee gpurun_out/r06_xcd_binding_ab.log
