#!/bin/bash
# A/B of the longest-first predictor (DMC_LPT_ALPHA) and of pipelined part-batches.
mkdir -p gpurun_out
{
for c in 4 3; do
  for v in "" lpt16 lpt3; do
    echo "== config $c variant '${v:-main(alpha5)}'"
    for rep in 1 2; do
      DMC_LIB_VARIANT=$v DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('rollout',{}))"
    done
    DMC_LIB_VARIANT=$v CONFIG=$c timeout 300 python scripts/occupancy_probe.py 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  try: d=json.loads(l)
  except Exception: continue
  print({k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('utilisation','last_start','max_item_over_span','corr_prev_cost_vs_dur','span_ticks')})
"
  done
done
for c in 4 3 5 2; do
  echo "== pipeline config $c"
  K=$([ $c = 2 ] && echo 1000 || echo 100) CONFIG=$c timeout 300 python scripts/pipeline_probe.py 2>&1 | tail -4
done
} > gpurun_out/lpt_ab.log 2>&1
tail -80 gpurun_out/lpt_ab.log
