#!/bin/bash
# Stages the reference's Python sources and XML assets (no meshes / mocap data: ~4 MB) in _refstage/ so that a gpurun
# snapshot carries them to the GPU box, where /root/reference does not exist (tests/ref_root.py).  _refstage/ is
# git-ignored: nothing of the reference is committed.
set -e
cd "$(dirname "$0")/.."
rm -rf _refstage
mkdir -p _refstage
(cd /root/reference && find dm_control \( -name '*.py' -o -name '*.xml' -o -name 'test_00*.textproto' -o -path '*soccer/assets/boxhead/*.png' -o -path '*soccer/assets/pitch/*.png' -o -path '*soccer/assets/soccer_ball/*.png' -o -path '*walkers/assets/jumping_ball/*.png' -o -path '*mujoco/testing/assets/*.stl' -o -path '*mujoco/testing/assets/deepmind.png' \) -size -600k -print0 | tar --null -T - -cf -) | tar -xf - -C _refstage
du -sh _refstage
