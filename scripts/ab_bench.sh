#!/bin/bash
# A/B of library variants on ONE box: VARIANTS="a b" (the empty name "-" is the production library), CONFIG=2..5
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-3}); do
  for v in ${VARIANTS:-- noentry}; do
    if [ "$v" = "-" ]; then unset DMC_LIB_VARIANT; else export DMC_LIB_VARIANT=$v; fi
    python bench.py --config ${CONFIG:-2} --no-cpu-baseline --parity-steps 0 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant ${v} rep ${rep}: value %.4g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)))"
  done
done
