#!/bin/bash
# Round 6: matrix-core instruction counters of the config-4 launch (chol_factor_tiles) -- rocprofv3 --pmc in its own pass.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$(pwd)
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_MFMA" | sort -u > $R/gpurun_out/r06_mfma_counters_avail.txt
cat $R/gpurun_out/r06_mfma_counters_avail.txt | tr '\n' ' '; echo
for c in 4 2; do
  DMC_BENCH_NO_PMC=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES -d $R/gpurun_out/r06_mfma_pmc_cfg$c -o p --output-format csv -- python $R/bench.py --pmc-child --config $c --steps 10 --warmup 5 > /dev/null 2> $R/gpurun_out/r06_mfma_pmc_cfg$c.err; echo "cfg $c rc=$?"
  python - <<PY
import csv, glob, statistics
per = {}
for f in glob.glob('$R/gpurun_out/r06_mfma_pmc_cfg$c/**/*counter_collection.csv', recursive=True):
  for row in csv.DictReader(open(f)):
    if 'step_kernel' in row.get('Kernel_Name', ''):
      d = per.setdefault(int(row['Dispatch_Id']), {})
      d[row['Counter_Name']] = d.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
ids = sorted(per)[-10:]
names = sorted({k for i in ids for k in per[i]})
print('config $c, median over the last', len(ids), 'step_kernel launches:', {n: statistics.median(per[i].get(n, 0.0) for i in ids) for n in names})
PY
done 2>&1 | tee $R/gpurun_out/r06_mfma_pmc.log
