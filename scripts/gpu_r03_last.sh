#!/bin/bash
# last call of the round (60 s of budget): the large models must run their max-ilp kernels again (dispatch fixed),
# then their fp32 parity test against the oracle
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for c in 4 5; do
  timeout 25 python bench.py --config $c --no-cpu-baseline --parity-steps 0 2>/dev/null > gpurun_out/r03_last_cfg$c.json
  python -c "
import json
d=json.load(open('gpurun_out/r03_last_cfg$c.json')); print('cfg$c value %.4g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)))"
done
timeout 40 python -m pytest tests/test_gpu_suite.py -m gpu -x -q -k "fp32_error_of_one_physics_step" 2>&1 | tail -2
