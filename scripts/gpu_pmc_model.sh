#!/bin/bash
# SQ / instruction-cache counters of the step kernel for one model: gpu_pmc_model.sh MODEL NSUB
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); M=$1; export MODEL=$1 NSUB=$2 REPS=${3:-20}
cd /tmp
[ -f $R/gpurun_out/counters_list.txt ] || rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
grep -o -i -E "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_CACHE)[A-Z_]*)\b" $R/gpurun_out/counters_list.txt | sort -u | tr '\n' ' '; echo
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
P3="SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/pmc_${M}_$i -o p --output-format csv -- python $R/scripts/step_target.py > $R/gpurun_out/pmc_${M}_$i.log 2>&1; echo "pmc $M pass $i rc=$?"
done
cd $R
python - <<PY
import csv, glob, statistics, collections, json
out = {}
for i in (1, 2, 3):
  for f in glob.glob('gpurun_out/pmc_${M}_%d/**/*counter_collection.csv' % i, recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      if 'step_kernel' in r.get('Kernel_Name', ''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
      v = v[-int('$REPS'):]
      out[k] = statistics.median(v); print('pass', i, k, 'median', statistics.median(v), 'n', len(v))
  for f in glob.glob('gpurun_out/pmc_${M}_%d/**/*kernel_trace.csv' % i, recursive=True):
    d = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'step_kernel' in r.get('Kernel_Name', '')]
    d = d[-int('$REPS'):]
    out['kernel_us_pass%d' % i] = statistics.median(d) / 1e3; print('pass', i, 'kernel us median', statistics.median(d) / 1e3)
json.dump(out, open('gpurun_out/pmc_${M}.json', 'w'), indent=1)
PY
tail -2 gpurun_out/pmc_${M}_3.log
