#!/bin/bash
# After splitting the fp32 large-model kernels into their own max-ilp-scheduled unit: GPU tests, the bench lines of
# configs 4 and 5 (the ones the split changes), the host-buffer (PCIe-inclusive) probe of config 2.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in 4 5; do
  timeout 100 python bench.py --config $c > gpurun_out/r03_bench_cfg$c.json 2> gpurun_out/r03_bench_cfg$c.err; echo "bench cfg$c rc=$?"
  python -c "
import json
d=json.load(open('gpurun_out/r03_bench_cfg$c.json')); print('cfg$c', d['value'], d['ms_per_step'], d['roofline'].get('traffic_over_algorithmic'), d.get('roofline_issue',{}).get('frac'))"
done
CONFIG=2 timeout 60 python scripts/pcie_probe.py 2>&1 | tail -1
