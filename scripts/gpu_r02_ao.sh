#!/bin/bash
# Round-2 GPU session AO: warm-start evaluation reused as the solver's starting point vs the previous library
# (libdmc_hip_p13.so), four configs on ONE box; full GPU tests
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('AB cfg$c $label', round(d['value']), round(d['rollout']['value']), sum(d['warnings_after_run']))"
}
for c in 3 4 5 2; do
  run p13 $c DMC_LIB_VARIANT=p13
  run new $c DMC_X=0
  run p13 $c DMC_LIB_VARIANT=p13
  run new $c DMC_X=0
done
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_ao.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed\|^FAILED\|^E  " gpurun_out/pytest_gpu_ao.log | head -12
