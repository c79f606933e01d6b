#!/bin/bash
# Round-2 GPU session AM: library with the CG solver vs the previous commit's (libdmc_hip_prev.so) on the four configs, ONE
# box (the model-specialised Newton kernels must be unchanged); CG GPU tests
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('AB cfg$c $label', round(d['value']), round(d['rollout']['value']), sum(d['warnings_after_run']))"
}
for c in 4 2 3 5; do
  run prev $c DMC_LIB_VARIANT=prev
  run new $c DMC_X=0
  run prev $c DMC_LIB_VARIANT=prev
  run new $c DMC_X=0
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -rP -k "cg_solver" > gpurun_out/pytest_gpu_am.log 2>&1; echo "pytest rc=$?"
grep -a "measured:\| passed\| failed\|^E  " gpurun_out/pytest_gpu_am.log | head -12
