#!/bin/bash
# Round-2 GPU session G: stash tests + finer solver phase profiles
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stash or equalities" > gpurun_out/pytest_gpu_g.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu_g.log
for mn in humanoid:5 cmu_2019_position_floor:6 soccer_2v2_boxhead:5; do
  MODEL=${mn%%:*} NSUB=${mn##*:} B=4096 timeout 600 python scripts/phase_profile_model.py > gpurun_out/phase_${mn%%:*}.log 2>&1; echo "phase rc=$?"; cat gpurun_out/phase_${mn%%:*}.log | cut -c1-300
done
