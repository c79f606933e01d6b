#!/bin/bash
# Round-2 GPU session A: gpu tests, smoke, bench, probes of the other BASELINE models.  Outputs -> gpurun_out/
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
MODELS=humanoid,cheetah timeout 600 python scripts/model_probe.py > gpurun_out/model_probe.log 2>&1; echo "probe rc=$?"; grep -v "^\s*$" gpurun_out/model_probe.log | cut -c1-400 | tail -12
timeout 900 python scripts/cmu_probe.py > gpurun_out/cmu_probe.log 2>&1; echo "cmu rc=$?"; cut -c1-600 gpurun_out/cmu_probe.log | tail -8
