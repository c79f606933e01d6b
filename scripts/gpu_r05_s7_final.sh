#!/bin/bash
# Closing session of round 5 (session 7): A/B of the final library against the library the session started from
# (libdmc_hip_base.so = commit 5ccff94) on this box, then the round's closing measurement script.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
CFGS="2 3 4 5" REPS=2 bash scripts/gpu_r05_s7b.sh > /dev/null 2>&1; cp gpurun_out/s7b.log gpurun_out/r05_ab_final_vs_session_start.log; cat gpurun_out/s7b.log
bash scripts/gpu_r05_final.sh
