#!/bin/bash
# Round 6, last session, call A: phase profiles of configs 2 / 3 / 5 / 4 with the tile-factorisation library (profiling build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for c in ${CFGS:-2 5 3 4}; do CONFIG=$c timeout 300 python scripts/phase_profile_cfg.py 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r06_s7_phase.log 2>&1
cat gpurun_out/r06_s7_phase.log | grep -v "^{" 
