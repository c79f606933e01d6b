#!/bin/bash
# Round-2 GPU session B: per-phase cycle profiles (profiling build) of cheetah, humanoid, CMU floor, soccer
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in cheetah:1 humanoid:5 cmu_2019_position_floor:6 soccer_2v2_boxhead:5; do
  M=${spec%%:*}; N=${spec##*:}
  MODEL=$M NSUB=$N timeout 600 python scripts/phase_profile_model.py > gpurun_out/phase_$M.log 2>&1; echo "$M rc=$?"; grep -v amdgpu.ids gpurun_out/phase_$M.log | tail -24
done
