#!/bin/bash
# HBM traffic of the bench launch: FETCH_SIZE / WRITE_SIZE in separate PMC passes (csv) -> gpurun_out/pmc_*
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --parity-steps 0 > /dev/null 2>> $R/gpurun_out/pmc.err; echo "pmc $c rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, statistics
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
  for f in glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'step_kernel' in r.get('Kernel_Name', '') and r.get('Counter_Name') == c]
    v = [float(r['Counter_Value']) for r in rows]
    if v: print(c, 'launches', len(v), 'median KB', statistics.median(v), 'mean', sum(v)/len(v))
PY
