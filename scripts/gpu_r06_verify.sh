#!/bin/bash
# Round 6, last check of the committed tree on one box: the whole GPU tier (serial, as the driver runs it), smoke(), the
# driver's bench command.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -m gpu -x -q -rs 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r06_verify_gputests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r06_verify_smoke.log
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_verify_bench_driver_cmd.json 2>/dev/null ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/r06_verify_bench_driver_cmd.json'))
print('value', round(d['value']), {k: round(v['value']) for k, v in d['extra']['configs'].items()}, 'env_step', round(d['extra']['env_step']['value']), d['roofline']['frac'], d['cpu_baseline']['value'])"
