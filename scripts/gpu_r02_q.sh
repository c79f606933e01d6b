#!/bin/bash
# Round-2 GPU session Q: A/B on one box -- M v through the tree vs the sparse row walk (libdmc_hip_old.so = previous commit)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_gpu_q.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_q.log
for rep in 1 2; do
for v in default old; do
  if [ $v = default ]; then unset DMC_LIB_VARIANT; else export DMC_LIB_VARIANT=$v; fi
  for c in 4 5; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 --steps 40 > gpurun_out/bench_q_${v}_$c.json 2> gpurun_out/bench_q_${v}_$c.err
    python -c "
import json; d=json.load(open('gpurun_out/bench_q_${v}_$c.json')); print('$v rep $rep cfg$c', round(d['value']), round(d['ms_per_step'],4), round(d['rollout']['value']), d['warnings_after_run'])"
  done
done; done
unset DMC_LIB_VARIANT
