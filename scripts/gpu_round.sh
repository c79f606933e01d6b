#!/bin/bash
# One GPU session: gpu tests, smoke, bench, rocprof kernel stats.  Outputs -> gpurun_out/
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "rocprof rc=$?"
ls -R gpurun_out/prof | head -20
