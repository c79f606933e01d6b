#!/bin/bash
# Round-2 GPU session R: HIP-graph capture of the composer control step; composer runs eager vs graph
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_composer.py tests/test_gpu_suite.py -x -q -m gpu > gpurun_out/pytest_gpu_r.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu_r.log
GRAPH=1 T=300 timeout 900 python scripts/composer_runs.py > gpurun_out/composer_runs.log 2>&1; echo "composer rc=$?"; cut -c1-330 gpurun_out/composer_runs.log | tail -12
