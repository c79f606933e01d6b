#!/bin/bash
# round 5, first GPU session: the (MjModel, MjData) seam on libdmc_hip.so -- the reference's own engine / core / index /
# thread-safety / lqr unit tests, and the seam's own tests -- then a baseline bench line per config for this box.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mujoco_api.py tests/test_reference_mujoco.py -m gpu -q -x --timeout 600 > gpurun_out/r05_seam_tests.log 2>&1
echo "seam tests rc=$?" | tee -a gpurun_out/r05_seam_tests.log
tail -5 gpurun_out/r05_seam_tests.log
for cfg in 2 5; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 > gpurun_out/r05_base_cfg$cfg.json 2> gpurun_out/r05_base_cfg$cfg.err
  python -c "
import json; d=json.loads(open('gpurun_out/r05_base_cfg$cfg.json').read()); print('cfg$cfg', d['value'], d['ms_per_step'])"
done
