#!/bin/bash
# Round-2 GPU session AN: full GPU test-suite and smoke with the final library
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -rP > gpurun_out/pytest_gpu_an.log 2>&1; echo "pytest rc=$?"
grep -a "measured:\| passed\| failed\|^FAILED\|^E  " gpurun_out/pytest_gpu_an.log | head -24
timeout 600 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('bench default', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline_issue']['frac'], d['cpu_baseline']['value'], d['parity']['open-loop']['max'])"
