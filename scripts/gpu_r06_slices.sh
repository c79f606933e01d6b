#!/bin/bash
# Round 6: sliced items of queued launches (StepIO::slices) -- bit-equality tests, then A/B against whole items on one box.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_suite.py -m gpu -q -x -k "work_queue or sliced" 2>&1 | tail -5
for rep in 1 2; do for c in 4 3; do for sl in 1 2 3 8; do
  DMC_SLICES=$sl DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 --pipeline 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg $c slices $sl value %.5g ms %.4f rollout %.5g' % (d['value'], d['ms_per_step'], d['rollout']['value']))"
done; done; done 2>&1 | tee gpurun_out/r06_slices_ab.log
for c in 4 3; do CONFIG=$c timeout 300 python scripts/queue_probe.py | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], d['ms_launch']); [print(json.dumps({k:v for k,v in l.items() if not isinstance(v,list)})) for l in d['launches'][-1:]]"; done
