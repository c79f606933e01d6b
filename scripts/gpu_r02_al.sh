#!/bin/bash
# Round-2 GPU session AL: CG solver on the device (generic kernel) -- full GPU tests; bench of the four configs to
# confirm the model-specialised Newton kernels are unchanged
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_al.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed" gpurun_out/pytest_gpu_al.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_al.log | head -20; grep -a "^E  " gpurun_out/pytest_gpu_al.log | head -10
for c in 2 3 4 5; do
  timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg$c', round(d['value']), round(d['rollout']['value']), sum(d['warnings_after_run']))"
done
