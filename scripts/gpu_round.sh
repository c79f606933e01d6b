#!/bin/bash
# One GPU session: gpu tests, smoke, bench, rocprof kernel stats + HBM PMC passes.  Outputs -> gpurun_out/
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
timeout 900 python bench.py --precision 64 --no-cpu-baseline > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err; echo "bench f64 rc=$?"; cat gpurun_out/bench_f64.json
R=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err; echo "rocprof rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/prof.err; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/prof.err; echo "pmc write rc=$?"
ls $R/gpurun_out/prof $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write | head -20
