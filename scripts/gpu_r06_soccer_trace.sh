#!/bin/bash
# soccer_2v2 as an environment at B = 256: what a control step is made of, with the task layer as two kernels
# (tasks/soccer_task.hip) and as tensor operations -- rocprofv3 kernel trace of scripts/soccer_env_trace.py
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
L=gpurun_out/r06_soccer_task_kernels.log; : > $L
for K in 1 0; do
  echo "== task_kernels=$K" >> $L
  KERNELS=$K python scripts/soccer_env_trace.py >> $L
  rm -rf /tmp/prof; KERNELS=$K rocprofv3 --kernel-trace --stats -d /tmp/prof -o s --output-format csv -- python scripts/soccer_env_trace.py > /dev/null 2>&1
  python - >> $L <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/prof/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
steps = [i for i, r in enumerate(rows) if 'step_kernel' in r['Kernel_Name']]
lo, hi = steps[len(steps) // 4], steps[3 * len(steps) // 4]
seg = rows[lo:hi]; n = sum('step_kernel' in r['Kernel_Name'] for r in seg)
wall = int(rows[hi]['Start_Timestamp']) - int(seg[0]['Start_Timestamp'])
phys = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg if 'step_kernel' in r['Kernel_Name'])
print('under rocprofv3, %d control steps: %.1f launches per control step, %.1f us per control step, of which the physics launch %.1f us = %.3f'
      % (n, len(seg) / n, wall / n / 1e3, phys / n / 1e3, phys / wall))
t0 = int(seg[0]['Start_Timestamp'])
for r in seg[:min(len(seg), int(len(seg) / n) + 1)]:
  print('%8.1f us +%7.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:100]))
PY
done
cat $L | cut -c1-160 | grep -v "^ .*at::native" | head -60
