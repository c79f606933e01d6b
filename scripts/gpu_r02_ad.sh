#!/bin/bash
# Round-2 GPU session AD: LDS bank-conflict and instruction-mix counters of the bench launches (configs 2, 3)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in 2 3; do
  K=$([ $c = 2 ] && echo 100 || echo 20)
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pmcL_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2> $R/gpurun_out/pmcL_cfg$c.err; echo "pmcL cfg $c rc=$?"; tail -2 $R/gpurun_out/pmcL_cfg$c.err
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH -d $R/gpurun_out/pmcM_cfg$c -o p --output-format csv -- python $R/bench.py --config $c --steps $K --warmup 5 --no-cpu-baseline --parity-steps 0 > /dev/null 2> $R/gpurun_out/pmcM_cfg$c.err; echo "pmcM cfg $c rc=$?"; tail -2 $R/gpurun_out/pmcM_cfg$c.err
done
cd $R
python - <<'PY'
import csv, glob, collections, statistics
for c in (2, 3):
  for tag in ('L', 'M'):
    acc = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pmc%s_cfg%d/**/*counter_collection.csv' % (tag, c), recursive=True):
      for r in csv.DictReader(open(f)):
        if 'step_kernel' in r.get('Kernel_Name', ''):
          acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print('cfg', c, tag, {k: statistics.median(v) for k, v in acc.items()})
PY
