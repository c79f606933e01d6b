#!/bin/bash
# Round 6, session 1: the round-5 library on this round's box -- bench lines of the four configs (no PMC / CPU legs),
# work-queue utilisation of configs 3 / 4 (scripts/queue_probe.py), wave tail of config 2.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
for c in 2 3 4 5; do
  DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/r06_s1_bench_cfg$c.json 2> gpurun_out/r06_s1_bench_cfg$c.err; echo "bench cfg$c rc=$?"
  python -c "
import json
d=json.load(open('gpurun_out/r06_s1_bench_cfg$c.json')); print('cfg$c', round(d['value']), d['ms_per_step'], 'rollout', round(d['rollout']['value']), 'pipe', round(d.get('pipelined',{}).get('value',0)))"
done
for c in 4 3; do CONFIG=$c timeout 300 python scripts/queue_probe.py > gpurun_out/r06_s1_queue_probe_cfg$c.log 2>&1; echo "queue probe cfg$c rc=$?"; done
timeout 100 python scripts/tail_probe.py > /dev/null 2>&1; cp gpurun_out/tail_probe_cheetah.json gpurun_out/r06_s1_wave_tail_cfg2.json
python - <<'PY'
import json
for c in (4, 3):
  try:
    d = json.load(open('gpurun_out/queue_probe_cfg%d.json' % c))
    print(c, d['ms_launch'], json.dumps(d['launches'][-1]))
  except Exception as e: print(c, e)
PY
