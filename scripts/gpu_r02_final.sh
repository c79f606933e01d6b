#!/bin/bash
# Round-2 final GPU session: full GPU test-suite, rocprofv3 kernel stats + PMC passes of the bench configs 2-5,
# profiles summarised ON the box (bench.py reads roofline.traffic / roofline_issue from them), then the bench lines
# with cpu_baseline and parity, phase profiles, composer runs; everything judged is copied to gpurun_out/profiles_new/
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out gpurun_out/profiles_new
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
cd /tmp
for c in 2 3 4 5; do
  K=$([ $c = 2 ] && echo 300 || echo 30)
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > $R/gpurun_out/prof_cfg$c.json 2> $R/gpurun_out/prof_cfg$c.err; echo "rocprof cfg $c rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcA_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc_cfg.err; echo "pmcA cfg $c rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmcB_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc_cfg.err; echo "pmcB cfg $c rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcF_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc_cfg.err; echo "pmcF cfg $c rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmcW_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc_cfg.err; echo "pmcW cfg $c rc=$?"
  cp $R/gpurun_out/prof_cfg$c/*kernel_stats.csv $R/gpurun_out/profiles_new/r02_rocprof_kernel_stats_cfg$c.csv 2>/dev/null || find $R/gpurun_out/prof_cfg$c -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/profiles_new/r02_rocprof_kernel_stats_cfg$c.csv \;
done
cd $R
python scripts/r02_profiles.py > gpurun_out/r02_profiles.log 2>&1; tail -3 gpurun_out/r02_profiles.log | cut -c1-300
for c in 2 3 4 5; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_cfg$c.json 2> gpurun_out/bench_cfg$c.err; echo "bench cfg $c rc=$?"; tail -3 gpurun_out/bench_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_cfg$c.json'))
print('cfg$c', 'value', d['value'], 'ms', d['ms_per_step'], 'phys/s', d['physics_steps_per_s'], 'rollout', d['rollout']['value'], 'parity', {k: (v['max'], v['median']) for k, v in d.get('parity', {}).items() if isinstance(v, dict)}, d.get('parity_error'), 'cpu', d.get('cpu_baseline'), 'warn', d['warnings_after_run'], d['workload_stats'], 'roof', d['roofline']['frac'], d['roofline'].get('traffic'), d.get('roofline_issue', {}).get('frac'), d['config']['info'])
PY
  cp gpurun_out/bench_cfg$c.json profiles/r02_bench_cfg$c.json
done
for mn in cheetah:1 humanoid:5 cmu_2019_position_floor:6 soccer_2v2_boxhead:5; do
  MODEL=${mn%%:*} NSUB=${mn##*:} B=4096 timeout 600 python scripts/phase_profile_model.py > gpurun_out/phase_${mn%%:*}.log 2>&1; echo "phase rc=$?"
  cp gpurun_out/phase_${mn%%:*}.json gpurun_out/profiles_new/r02_phase_${mn%%:*}.json
done
GRAPH=1 T=300 timeout 900 python scripts/composer_runs.py > gpurun_out/composer_runs.log 2>&1; echo "composer rc=$?"; cut -c1-330 gpurun_out/composer_runs.log | tail -8
T=300 timeout 1500 python scripts/soak.py > gpurun_out/soak.log 2>&1; echo "soak rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/soak.json'))
print('soak tasks', len(d), 'errors', [r for r in d if 'error' in r][:3], 'nonzero warnings', [(r['task'], r['warnings']) for r in d if 'warnings' in r and sum(r['warnings'])])
PY
cp gpurun_out/soak.json gpurun_out/profiles_new/r02_soak_all_tasks.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
cp profiles/r02_bench_cfg*.json profiles/r02_kernel_stats_cfg*.json profiles/r02_pmc_cfg*.json gpurun_out/profiles_new/
cp gpurun_out/composer_runs.json gpurun_out/profiles_new/r02_composer_runs.json 2>/dev/null
ls gpurun_out/profiles_new
