#!/bin/bash
# instruction mix of the bench launch (SQ counters, one pass) -> gpurun_out/pmc_inst
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_inst -o p --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc.err; echo "pmc inst rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $R/gpurun_out/pmc_inst2 -o p --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc.err; echo "pmc inst2 rc=$?"
cd $R
python - <<'PY'
import csv, glob, statistics, collections
for d in ('pmc_inst', 'pmc_inst2'):
  for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv' % d, recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      if 'step_kernel' in r.get('Kernel_Name', ''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(d, k, 'median', statistics.median(v), 'n', len(v))
PY
