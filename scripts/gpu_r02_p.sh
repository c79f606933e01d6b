#!/bin/bash
# Round-2 GPU session P: A/B on one box -- sub-stages out of line (DMC_SPLIT variants)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for v in default s16 s15 s63 s32; do
  if [ $v = default ]; then unset DMC_LIB_VARIANT; else export DMC_LIB_VARIANT=$v; fi
  for c in 2 3; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 --steps $([ $c = 2 ] && echo 1000 || echo 40) > gpurun_out/bench_p_${v}_$c.json 2> gpurun_out/bench_p_${v}_$c.err
    python -c "
import json; d=json.load(open('gpurun_out/bench_p_${v}_$c.json')); print('$v rep $rep cfg$c', round(d['value']), round(d['ms_per_step'],4), round(d['rollout']['value']))"
  done
done; done
unset DMC_LIB_VARIANT
