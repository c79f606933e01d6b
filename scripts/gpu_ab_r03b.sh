#!/bin/bash
# one call: GPU tests, then A/B of the production library against variants on configs 5 and 4
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for c in ${CONFIGS:-5 4}; do
  echo "== config $c"
  CONFIG=$c REPS=${REPS:-2} VARIANTS="${VARIANTS:-- nonsr}" bash scripts/ab_bench.sh
done
