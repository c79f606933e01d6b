#!/bin/bash
# the baked kernels of libdmc_hip_base.so (commit 5ccff94) against named plugins of the working tree, one box:  CFGS="3 4" REPS=2 bash scripts/gpu_r05_base_vs_plugin.sh <tag>
mkdir -p gpurun_out
{
for c in ${CFGS:-3}; do for rep in $(seq ${REPS:-2}); do
  DMC_LIB_VARIANT=base DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 --pipeline 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg $c base        ', 'value %.5g ms %.4f rollout %.5g iter %.3f' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter']))"
  for t in "$@"; do CFG=$c REPS=1 bash scripts/gpu_r05_plugins.sh $t | tail -1; done
done; done
} > gpurun_out/base_vs_plugin.log 2>&1
cat gpurun_out/base_vs_plugin.log
