#!/bin/bash
# Round-2 GPU session I: full GPU test-suite (incl. composer layer), composer environment runs
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_composer.py -x -q -m gpu > gpurun_out/pytest_gpu_i.log 2>&1; echo "pytest composer rc=$?"
tail -30 gpurun_out/pytest_gpu_i.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/pytest_gpu.log
T=${T:-300} timeout 900 python scripts/composer_runs.py > gpurun_out/composer_runs.log 2>&1; echo "composer rc=$?"; cut -c1-700 gpurun_out/composer_runs.log | tail -8
