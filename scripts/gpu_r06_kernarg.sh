#!/bin/bash
# Round 6: HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory instead of host-coherent memory) on the launch-per-step
# loops: config 2 (66 us launches) and the fused suite environment, one box.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2 3; do for k in 0 1; do for c in 2 5; do
  HIP_FORCE_DEV_KERNARG=$k DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('cfg $c rep $rep HIP_FORCE_DEV_KERNARG=$k value %.5g ms %.5f kernel_ms %.5f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg']))"
done; done; done
for k in 0 1; do
  HIP_FORCE_DEV_KERNARG=$k DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('driver cmd HIP_FORCE_DEV_KERNARG=$k value %.5g env_step %.5g' % (d['value'], d['extra']['env_step']['value']))"
done
} 2>&1 | tee gpurun_out/r06_kernarg_ab.log
