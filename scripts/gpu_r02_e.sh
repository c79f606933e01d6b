#!/bin/bash
# Round-2 GPU session E: tests + bench with / without the HBM stash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
for v in stash nostash; do
  if [ $v = nostash ]; then export DMC_NO_STASH=1; else unset DMC_NO_STASH; fi
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['rollout']['value'], d['max_rel_qpos_err_vs_cpu'], d['config']['info'])"
done
unset DMC_NO_STASH
MODELS=humanoid,walker,hopper,cartpole timeout 600 python scripts/model_probe.py > gpurun_out/model_probe.log 2>&1; echo "probe rc=$?"; grep -v "^\s*$" gpurun_out/model_probe.log | cut -c1-400 | tail -12
