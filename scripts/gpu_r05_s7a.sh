#!/bin/bash
# session 7, call A: phase profiles of configs 2 and 5 with the library of commit 5ccff94 (profiling build), wave tail of config 2
mkdir -p gpurun_out
{
for c in 2 5; do CONFIG=$c timeout 300 python scripts/phase_profile_cfg.py 2>&1 | grep -v amdgpu.ids; done
CONFIG=2 timeout 300 python scripts/tail_probe.py 2>&1 | tail -30
} > gpurun_out/s7a.log 2>&1
cat gpurun_out/s7a.log
