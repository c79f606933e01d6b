#!/bin/bash
# 62 x 62 fp32 Cholesky: row-per-lane (production form) vs tiles on the matrix cores (scripts/chol_mfma_probe.hip)
cd /root/repo; mkdir -p gpurun_out
for V in 1 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBCAST_BPERMUTE=$V -o /tmp/chol_probe_$V scripts/chol_mfma_probe.hip 2>/dev/null || exit 1
  echo "== row-group broadcast by $([ $V = 1 ] && echo ds_bpermute || echo "permlane swaps on the diagonal tile, ds_bpermute on the rest")"; /tmp/chol_probe_$V
done 2>&1 | tee gpurun_out/r06_chol_mfma_probe.log
