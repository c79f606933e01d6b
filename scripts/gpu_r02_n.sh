#!/bin/bash
# Round-2 GPU session N: full tests; the N > 1 bench path on one device (gloo, both ranks on cuda:0); composer runs
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log
for c in 2 4; do
  DMC_BENCH_BACKEND=gloo DMC_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config $c --steps 20 --warmup 3 --batch 1024 --parity-steps 5 > gpurun_out/bench_2rank_cfg$c.json 2> gpurun_out/bench_2rank_cfg$c.err; echo "2-rank bench cfg $c rc=$?"; tail -2 gpurun_out/bench_2rank_cfg$c.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/bench_2rank_cfg$c.json') if l.startswith('{')][-1])
print('2rank cfg$c', d['n_gpus'], d['value'], d['ms_per_step'], d.get('collectives'), d['config']['sharding'])
PY
done
T=300 timeout 900 python scripts/composer_runs.py > gpurun_out/composer_runs.log 2>&1; echo "composer rc=$?"; cut -c1-420 gpurun_out/composer_runs.log | tail -6
DOMAINS=manipulator T=300 timeout 600 python scripts/soak.py > gpurun_out/soak_n.log 2>&1; grep "^{" gpurun_out/soak_n.log | cut -c1-220
