#!/bin/bash
# Round-2 GPU session F: tests + per-phase cycle profiles of the BASELINE models
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for mn in cheetah:1 humanoid:5 cmu_2019_position_floor:6 soccer_2v2_boxhead:5; do
  MODEL=${mn%%:*} NSUB=${mn##*:} B=4096 timeout 600 python scripts/phase_profile_model.py > gpurun_out/phase_${mn%%:*}.log 2>&1; echo "phase rc=$?"; cat gpurun_out/phase_${mn%%:*}.log | cut -c1-300
done
