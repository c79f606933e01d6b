#!/bin/bash
# Round-3 measurement session: bench lines of configs 2-5 (parity legs, CPU baseline, live PMC passes), rocprofv3
# kernel stats of the default bench command, wave-tail probes.  Outputs -> gpurun_out/r03_*
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
for c in 2 3 4 5; do
  timeout 900 python bench.py --config $c > gpurun_out/r03_bench_cfg$c.json 2> gpurun_out/r03_bench_cfg$c.err; echo "bench cfg$c rc=$?"
  python -c "
import json
d=json.load(open('gpurun_out/r03_bench_cfg$c.json')); print('cfg$c', d['value'], d['ms_per_step'], d['roofline'].get('traffic_over_algorithmic'), d.get('roofline_issue',{}).get('frac'))"
done
R=$(pwd); cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof -o r03 --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > $R/gpurun_out/r03_prof_bench.json 2> $R/gpurun_out/r03_prof.err; echo "rocprof rc=$?"
cd $R
find gpurun_out/r03_prof -name "*stats*" | head; 
python scripts/tail_probe.py > gpurun_out/r03_tail_cheetah.log 2>&1
CONFIG=5 NOSLIP0=1 python scripts/tail_probe_cfg.py > /dev/null 2>&1
CONFIG=4 python scripts/tail_probe_cfg.py > /dev/null 2>&1
CONFIG=3 python scripts/tail_probe_cfg.py > /dev/null 2>&1
ls gpurun_out | head -40
