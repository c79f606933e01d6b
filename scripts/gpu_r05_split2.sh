mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_tree_split.py -q -m gpu -x 2>&1 | tail -4
for v in "" nosplit; do for rep in 1 2; do DMC_LIB_VARIANT=$v DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config 5 --no-cpu-baseline --parity-steps 0 --pipeline 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg 5', '$v', d['value'], d['ms_per_step'], d['rollout']['value'])"; done; done
} > gpurun_out/split2.log 2>&1
cat gpurun_out/split2.log
