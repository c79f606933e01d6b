#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "precision|passed|failed|Error|error" gpurun_out/pytest_gpu.log | tail -25
TAG=${TAG:-b} timeout 900 python scripts/perf_probe.py > gpurun_out/perf_probe_${TAG:-b}.log 2>&1; cat gpurun_out/perf_probe_${TAG:-b}.log | python -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    i=r.get('info',{})
    print('prec',r.get('prec'),'lanes',r.get('lanes'),'nstep',r.get('nstep'),'ms %.4f'%r.get('ms_per_launch',-1),'Msteps/s %.2f'%(r.get('steps_per_s',0)/1e6),'epb',i.get('envs_per_block'),'lds',i.get('lds_bytes_per_block'),'grid',i.get('grid'),'caps',i.get('nconmax'),i.get('njmax'), {k:r[k] for k in ('max_ncon','max_nefc','max_iter') if k in r}, r.get('error',''))
"
