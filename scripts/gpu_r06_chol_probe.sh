#!/bin/bash
# fp32 Cholesky of 33 .. 64-dof Hessians: row-per-lane against tiles on the matrix cores (scripts/chol_mfma_probe.hip builds
# the production routines of step_core.h); FLAGS="-D..." for a variant
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo; mkdir -p gpurun_out
for F in "${@:-}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result $F -o /tmp/chol_probe scripts/chol_mfma_probe.hip 2>/dev/null || exit 1
  echo "== flags '$F'"; /tmp/chol_probe | grep -E "N (48|62)"
done 2>&1 | tee gpurun_out/r06_chol_mfma_probe.log
