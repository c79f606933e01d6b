#!/bin/bash
# Round-5 closing measurement session, one box, final library: the GPU test tier (reference tree staged:
# scripts/stage_reference.sh), smoke, bench lines of configs 5, 4, 3, 2 (parity legs, CPU baseline, live PMC passes), fp64
# lines, the driver's command line, rocprofv3 kernel stats of every config, composer and suite device environments, wave
# tails.  Outputs -> gpurun_out/r05_* (scripts/r05_profiles.py copies the summaries into profiles/).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests -m gpu -q -rs -n 4 2>&1 | tail -8 > gpurun_out/r05_gputests.log; tail -2 gpurun_out/r05_gputests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; tail -2 gpurun_out/r05_smoke.log
for c in 5 4 3 2; do
  timeout 300 python bench.py --config $c > gpurun_out/r05_bench_cfg$c.json 2> gpurun_out/r05_bench_cfg$c.err; echo "bench cfg$c rc=$?"
  python -c "
import json
d=json.load(open('gpurun_out/r05_bench_cfg$c.json')); print('cfg$c', round(d['value']), d['ms_per_step'], d['roofline'].get('traffic_over_algorithmic'), d.get('roofline_issue',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))"
  timeout 200 python bench.py --config $c --precision 64 --no-cpu-baseline --parity-steps 0 > gpurun_out/r05_bench_f64_cfg$c.json 2>/dev/null
done
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd.json 2>/dev/null; echo "driver cmd rc=$?"
cd /tmp
for c in 2 3 4 5; do
  K=200; [ $c != 2 ] && K=30
  DMC_BENCH_NO_PMC=1 timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_prof_cfg$c -o r05 --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > $R/gpurun_out/r05_prof_bench_cfg$c.json 2> $R/gpurun_out/r05_prof_cfg$c.err; echo "rocprof cfg$c rc=$?"
done
cd $R
GRAPH=1 T=300 timeout 400 python scripts/composer_runs.py > gpurun_out/r05_composer.log 2>&1; echo "composer rc=$?"; cp gpurun_out/composer_runs.json gpurun_out/r05_composer_runs.json
timeout 900 python scripts/device_env_runs.py > gpurun_out/r05_device_env_runs.log 2>&1; tail -1 gpurun_out/r05_device_env_runs.log
timeout 100 python scripts/tail_probe.py > /dev/null 2>&1; cp gpurun_out/tail_probe_cheetah.json gpurun_out/r05_wave_tail_cfg2.json
for c in 3 4 5; do CONFIG=$c timeout 200 python scripts/tail_probe_cfg.py > /dev/null 2>&1; cp gpurun_out/tail_probe_cfg$c.json gpurun_out/r05_wave_tail_cfg$c.json; done
python scripts/r05_profiles.py
