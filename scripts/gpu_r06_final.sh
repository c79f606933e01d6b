#!/bin/bash
# Round-6 closing measurement session, one box, final library: the GPU test tier (reference tree staged:
# scripts/stage_reference.sh), smoke, bench lines of configs 5, 4, 3, 2 (parity legs, CPU baseline, live PMC passes), fp64
# lines, the driver's command line (with the extra legs), rocprofv3 kernel stats of every config, composer environments,
# the suite's environments through suite/fused_env.py (task layer inside the step kernel), wave tails, the work queue with
# whole items (what the sliced queue replaced).  Outputs -> gpurun_out/r06_* (scripts/r06_profiles.py -> profiles/).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q -rs -n 4 2>&1 | tail -8 > gpurun_out/r06_gputests.log; tail -2 gpurun_out/r06_gputests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -2 gpurun_out/r06_smoke.log
for c in 5 4 3 2; do
  timeout 400 python bench.py --config $c --extra 0 > gpurun_out/r06_bench_cfg$c.json 2> gpurun_out/r06_bench_cfg$c.err; echo "bench cfg$c rc=$?"
  python -c "
import json
d=json.load(open('gpurun_out/r06_bench_cfg$c.json')); print('cfg$c', round(d['value']), d['ms_per_step'], d['roofline'].get('traffic_over_algorithmic'), d.get('roofline_issue',{}).get('frac'), d.get('cpu_baseline',{}).get('value'), 'pipe', d.get('pipelined',{}).get('by_parts'), 'rollout', round(d['rollout']['value']))"
  timeout 200 python bench.py --config $c --precision 64 --no-cpu-baseline --parity-steps 0 --pipeline 0 --extra 0 > gpurun_out/r06_bench_f64_cfg$c.json 2>/dev/null
done
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.json 2>/dev/null ) 2>&1 | grep real; echo "driver cmd done"
cd /tmp
for c in 2 3 4 5; do
  K=200; [ $c != 2 ] && K=30
  DMC_BENCH_NO_PMC=1 timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_prof_cfg$c -o r06 --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 --pipeline 0 --extra 0 > $R/gpurun_out/r06_prof_bench_cfg$c.json 2> $R/gpurun_out/r06_prof_cfg$c.err; echo "rocprof cfg$c rc=$?"
done
cd $R
GRAPH=1 T=300 timeout 400 python scripts/composer_runs.py > gpurun_out/r06_composer.log 2>&1; echo "composer rc=$?"; cp gpurun_out/composer_runs.json gpurun_out/r06_composer_runs.json
T=1000 timeout 1800 python scripts/fused_env_runs.py > gpurun_out/r06_fused_env_runs.log 2>&1; tail -1 gpurun_out/r06_fused_env_runs.log
timeout 100 python scripts/tail_probe.py > /dev/null 2>&1; cp gpurun_out/tail_probe_cheetah.json gpurun_out/r06_wave_tail_cfg2.json
CONFIG=5 timeout 200 python scripts/tail_probe_cfg.py > /dev/null 2>&1; cp gpurun_out/tail_probe_cfg5.json gpurun_out/r06_wave_tail_cfg5.json
for c in 4 3; do CONFIG=$c timeout 300 python scripts/queue_probe.py > /dev/null 2>&1; cp gpurun_out/queue_probe_cfg$c.json gpurun_out/r06_queue_probe_cfg$c.json; done
DMC_BENCH_SINGLE_DEVICE=1 DMC_BENCH_BACKEND=gloo DMC_BENCH_NO_PMC=1 timeout 600 python bench.py --gpus 2 --config 2 --steps 50 --warmup 5 --no-cpu-baseline --parity-steps 0 --pipeline 0 2>/dev/null | tail -1 > gpurun_out/r06_2rank_gloo_single_device.json; echo "2-rank harness rc=$?"
REPS=2 bash scripts/ab_vs_round5.sh > /dev/null 2>&1; tail -16 gpurun_out/r06_ab_vs_round5.log
python scripts/r06_profiles.py
