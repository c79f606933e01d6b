#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${TAG:-a} timeout 900 python scripts/perf_probe.py > gpurun_out/perf_probe_${TAG:-a}.log 2>&1; tail -20 gpurun_out/perf_probe_${TAG:-a}.log
if [ -n "$PMC" ]; then
  cd /tmp
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-40)
    CFGS='[[32,64,0,0,1,false]]' TAG=pmc timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$n --output-format csv -- python $GRAFT_REPO_ROOT/scripts/perf_probe.py > /dev/null 2>> $GRAFT_REPO_ROOT/gpurun_out/pmc.err
  done
  ls $GRAFT_REPO_ROOT/gpurun_out/pmc | head
fi
