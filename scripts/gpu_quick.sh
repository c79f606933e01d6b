#!/bin/bash
# quick GPU loop: parity tests (subset or all), perf probe, phase profile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
CFGS=${CFGS:-'[[32,64,0,0,1,false],[32,32,0,0,1,false],[32,16,0,0,1,false],[32,32,0,0,10,false],[64,32,0,0,1,false]]'} TAG=q timeout 600 python scripts/perf_probe.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    i=r.get('info',{})
    print('prec',r.get('prec'),'lanes',r.get('lanes'),'nstep',r.get('nstep'),'ms %.4f'%r.get('ms_per_launch',-1),'Msteps/s %.2f'%(r.get('steps_per_s',0)/1e6),'rollout %.2f'%(r.get('rollout_steps_per_s',0)/1e6),'epb',i.get('envs_per_block'),'static',i.get('static_id'), r.get('error',''))
"
if [ -n "$PHASES" ]; then timeout 600 python scripts/phase_profile.py 2>&1 | head -45; fi
