#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$(pwd); cd /tmp
for m in ${MODES:-eager graph}; do
  MODE=$m timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ctrace_$m -o ct --output-format csv -- python $R/scripts/composer_trace.py 2>&1 | grep "ms per"
  python - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob('$R/gpurun_out/ctrace_$m/**/*kernel_trace.csv',recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//2:]   # steady state
c=collections.Counter(); d=collections.Counter()
for r in rows: n=r['Kernel_Name'].split('(')[0][-60:]; c[n]+=1; d[n]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
span=(int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e3
print('$m: kernels', len(rows), 'span us', round(span), 'busy us', round(sum(d.values())), 'kernels per step ~', round(len(rows)/100,1))
for n,v in d.most_common(12): print('   %-62s n %5d total %8.0f us avg %6.1f'%(n,c[n],v,v/c[n]))
PY
done
