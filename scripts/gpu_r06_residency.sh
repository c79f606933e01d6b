#!/bin/bash
# Round 6: five against six resident 62-dof environments per CU (scripts/residency_probe.py), one box, two repetitions
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
{ for rep in 1 2; do for w in 5 6; do NCONMAX=${NCONMAX:-14} WAVES=$w timeout 600 python scripts/residency_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; done; done; } | tee gpurun_out/r06_residency_probe.log
