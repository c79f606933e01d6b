#!/bin/bash
# Round-3 closing measurement session (after the register line search of the general row types): composer GPU tests,
# bench lines of configs 5, 4, 2, 3 (stalest first: parity legs, CPU baseline, live PMC passes), the driver's command
# line, rocprofv3 kernel stats of the default bench command.  Outputs -> gpurun_out/r03_*
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_composer.py -m gpu -x -q 2>&1 | tail -3
for c in 5 4 2 3; do
  timeout 200 python bench.py --config $c > gpurun_out/r03_bench_cfg$c.json 2> gpurun_out/r03_bench_cfg$c.err; echo "bench cfg$c rc=$?"
  python -c "
import json
d=json.load(open('gpurun_out/r03_bench_cfg$c.json')); print('cfg$c', d['value'], d['ms_per_step'], d['roofline'].get('traffic_over_algorithmic'), d.get('roofline_issue',{}).get('frac'))"
done
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_cmd.json 2>/dev/null; echo "driver cmd rc=$?"
R=$(pwd); cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof -o r03 --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > $R/gpurun_out/r03_prof_bench.json 2> $R/gpurun_out/r03_prof.err; echo "rocprof rc=$?"
cd $R
find gpurun_out/r03_prof -name "*stats*" | head
