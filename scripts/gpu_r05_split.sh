#!/bin/bash
# tree-split factorisations: bit identity + rate vs the -DDMC_NO_TREE_SPLIT library, soccer parity tests, bench of config 5
mkdir -p gpurun_out
{
timeout 600 python scripts/tree_split_ab.py
echo "== B=4096"
B=4096 T=50 timeout 600 python scripts/tree_split_ab.py
echo "== gpu tests (soccer / composer / abi)"
timeout 900 python -m pytest tests -q -m gpu -x -k "soccer or config5 or cfg5 or box_piles or islands or noslip or elliptic" 2>&1 | tail -5
echo "== bench config 5"
for v in "" nosplit; do DMC_LIB_VARIANT=$v DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['rollout']['value'], d.get('pipelined',{}).get('value'), d.get('max_rel_qpos_err_vs_cpu'), d.get('parity',{}).get('one-step'))"; done
} > gpurun_out/split_ab.log 2>&1
tail -40 gpurun_out/split_ab.log
