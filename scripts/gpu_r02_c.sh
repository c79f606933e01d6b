#!/bin/bash
# Round-2 GPU session C: gpu tests + bench + probes + per-phase profiles in one call
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print(d['value'], d['ms_per_step'], d['rollout']['value'], d['max_rel_qpos_err_vs_cpu'], d['config']['info'])"
MODELS=humanoid timeout 600 python scripts/model_probe.py > gpurun_out/model_probe.log 2>&1; echo "probe rc=$?"; grep -v "^\s*$" gpurun_out/model_probe.log | cut -c1-330 | tail -6
timeout 900 python scripts/cmu_probe.py > gpurun_out/cmu_probe.log 2>&1; echo "cmu rc=$?"; cut -c1-330 gpurun_out/cmu_probe.log | tail -4
for spec in cheetah:1 humanoid:5 cmu_2019_position_floor:6 soccer_2v2_boxhead:5; do
  M=${spec%%:*}; N=${spec##*:}
  MODEL=$M NSUB=$N timeout 600 python scripts/phase_profile_model.py > gpurun_out/phase_$M.log 2>&1; echo "$M rc=$?"; grep -v amdgpu.ids gpurun_out/phase_$M.log | tail -22
done
