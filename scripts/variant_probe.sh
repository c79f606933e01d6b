#!/bin/bash
# perf of experimental library variants (DMC_LIB_VARIANT): VARIANTS="a b c" bash scripts/variant_probe.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for v in ${VARIANTS:-""}; do
  [ "$v" = "base" ] && v=""
  echo "=== variant '$v'"
  DMC_LIB_VARIANT=$v CFGS=${CFGS:-'[[32,32,0,0,1,false],[32,32,0,0,10,false]]'} TAG=v$v timeout 600 python scripts/perf_probe.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    i=r.get('info',{})
    print('prec',r.get('prec'),'lanes',r.get('lanes'),'nstep',r.get('nstep'),'ms %.4f'%r.get('ms_per_launch',-1),'Msteps/s %.2f'%(r.get('steps_per_s',0)/1e6),'epb',i.get('envs_per_block'),'lds',i.get('lds_bytes_per_block'),'grid',i.get('grid'), r.get('error',''))
"
done
