#!/bin/bash
# Round 6: the ceiling of a soccer match on several waves, measured: fences per step on the host build of
# the kernel core, fence / barrier prices on the device, the phase profile of the match -> profiles/r06_multiwave_prototype.log
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "== fences per physics step (host build of the kernel core, scripts/multiwave_probe.py)"
timeout 600 python scripts/multiwave_probe.py 2>/dev/null | tail -1
echo "== fence / barrier cost on the device (scripts/barrier_cost_probe.hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_cost_probe scripts/barrier_cost_probe.hip && /tmp/barrier_cost_probe
echo "== config 5 launch (5 physics steps, B = 256, one wave per CU)"
DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config 5 --no-cpu-baseline --parity-steps 0 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('value %.5g env-steps/s, %.4f ms per launch = %.1f us per physics step' % (d['value'], d['ms_per_step'], 1e3*d['ms_per_step']/5))"
} > gpurun_out/r06_multiwave_prototype.log 2>&1
cat gpurun_out/r06_multiwave_prototype.log
