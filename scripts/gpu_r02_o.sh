#!/bin/bash
# Round-2 GPU session O: A/B on one box -- small-model helpers inlined into the stage functions vs out of line
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_gpu_o.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_o.log
for rep in 1 2; do
for v in default noinl; do
  if [ $v = noinl ]; then export DMC_LIB_VARIANT=noinl; else unset DMC_LIB_VARIANT; fi
  timeout 300 python bench.py --no-cpu-baseline --parity-steps 0 > gpurun_out/bench_o_$v.json 2> gpurun_out/bench_o_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_o_$v.json')); print('$v rep $rep', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['rollout']['value'])"
  MODELS=walker,hopper,cartpole OUTNAME=model_probe_o_$v.json timeout 300 python scripts/model_probe.py 2>&1 | grep '"prec": 32' | python -c "
import sys, json
for l in sys.stdin:
  d=json.loads(l); print('   $v', d['model'], d['lanes'], d.get('env_steps_per_s'), d.get('rollout_env_steps_per_s'))"
done; done
unset DMC_LIB_VARIANT
