#!/bin/bash
# Round-2 GPU session K: full GPU test-suite, bench configs 2-5, rocprofv3 kernel stats + PMC passes of the bench
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
for c in 2 3 4 5; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_cfg$c.json 2> gpurun_out/bench_cfg$c.err; echo "bench cfg $c rc=$?"; tail -3 gpurun_out/bench_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_cfg$c.json'))
print('cfg$c', 'value', d['value'], 'ms', d['ms_per_step'], 'phys/s', d['physics_steps_per_s'], 'rollout', d['rollout']['value'], 'parity', d.get('parity'), d.get('parity_error'), 'cpu', d.get('cpu_baseline'), 'warn', d['warnings_after_run'], d['workload_stats'], 'frac', d['roofline']['frac'], d['config']['info'])
PY
done
cd /tmp
for c in 2 3 4 5; do
  K=$([ $c = 2 ] && echo 300 || echo 30)
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > $R/gpurun_out/prof_cfg$c.json 2> $R/gpurun_out/prof_cfg$c.err; echo "rocprof cfg $c rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcA_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc_cfg.err; echo "pmcA cfg $c rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmcB_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc_cfg.err; echo "pmcB cfg $c rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcF_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc_cfg.err; echo "pmcF cfg $c rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmcW_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc_cfg.err; echo "pmcW cfg $c rc=$?"
done
cd $R
python scripts/r02_profiles.py
