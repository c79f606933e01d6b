#!/bin/bash
# Config 2: what the kinematic stash buys and what it costs in HBM traffic (VERDICT r05 #5c) -- the same library with and
# without it (DMC_NO_KSTASH=1), live PMC passes, two repetitions on one box -> profiles/r06_kstash_ab.log
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do for ks in "" 1; do
  if [ -n "$ks" ]; then export DMC_NO_KSTASH=1; else unset DMC_NO_KSTASH; fi
  timeout 300 python bench.py --config 2 --no-cpu-baseline --parity-steps 0 --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; i=d.get('roofline_issue',{})
print('kinematic stash %s rep $rep: value %.5g env-steps/s, %.4f ms per launch, rollout %.5g; HBM traffic %.2f MB per launch = %.1f x algorithmic; VALU instructions per launch %.4g' % ('OFF' if '$ks' else 'on ', d['value'], d['ms_per_step'], d['rollout']['value'], (r['traffic'] or 0)/1e6, r['traffic_over_algorithmic'] or 0, i.get('valu_insts_per_launch', 0)))"
done; done
} 2>&1 | tee gpurun_out/r06_kstash_ab.log
