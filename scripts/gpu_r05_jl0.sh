mkdir -p gpurun_out
{
for c in 5 3; do for v in "" jl0; do for rep in 1 2; do DMC_LIB_VARIANT=$v DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 --pipeline 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg $c', '$v', d['value'], d['ms_per_step'], d['rollout']['value'], d['config']['info']['lds_bytes_per_block'], d['config']['info']['envs_per_cu'], d['config']['info']['static_id'])"; done; done; done
DMC_LIB_VARIANT=jl0 B=4096 T=50 timeout 300 python scripts/tree_split_ab.py child /tmp/x.npz && python -c "
import numpy as np; d=np.load('/tmp/x.npz'); print('jl0 B=4096 ms', d['ms'])"
} > gpurun_out/jl0_ab.log 2>&1
cat gpurun_out/jl0_ab.log
