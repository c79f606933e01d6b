#!/bin/bash
# Round-2 GPU session Y: per-phase cycles (profiling build) of the four BASELINE models after the LDS diet 2 / work queue
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for mn in cheetah:1 humanoid:5 cmu_2019_position_floor:6 soccer_2v2_boxhead:5; do
  MODEL=${mn%%:*} NSUB=${mn##*:} B=4096 timeout 600 python scripts/phase_profile_model.py > gpurun_out/phase_${mn%%:*}.log 2>&1; echo "phase rc=$?"; grep -v amdgpu.ids gpurun_out/phase_${mn%%:*}.log | cut -c1-400 | tail -28
done
