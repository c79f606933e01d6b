"""Writes the XML that the reference's OWN PyMJCF composition code produces for BASELINE configs 4 and 5
(mjcf/element.py:817 `to_xml_string`, recompiled by composer/environment.py:377-383).

The reference sources are executed unmodified from /root/reference through tests/reference_pymjcf.py (which documents
the import seams: lxml -> xml.etree, absl, dm_env, no engine).  Only runnable where the reference tree exists; the
outputs are committed so that the GPU box and the judge's checkout can use them.

  tests/golden/pymjcf_cmu2019_go_to_target.xml     locomotion/examples/basic_cmu_2019.py:97-118: CMUHumanoidPositionControlled
                                                   + Floor + GoToTarget(physics_timestep=0.005, control_timestep=0.03)
  tests/golden/pymjcf_soccer_2v2_boxhead_seed0.xml locomotion/soccer/__init__.py:92-148 load(team_size=2): the model of the first
                                                   episode for random_state = RandomState(0) (task.initialize_episode_mjcf
                                                   draws the pitch size: soccer/pitch.py:663-676)
  dm_control_amd/suite/assets/soccer_2v2_boxhead.xml  the same composition with the pitch-size randomizer at its midpoint
                                                   (randomizer = 0.5 -> 40 x 30 half-extents): the BASELINE config-5 workload;
                                                   per-environment pitch sizes are applied on the device as geom deltas
                                                   (composer/tasks/soccer.py)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import reference_pymjcf as rp      # noqa: E402


def outputs():
  out = {}
  task = rp.cmu2019_go_to_target()
  out['tests/golden/pymjcf_cmu2019_go_to_target.xml'] = task.root_entity.mjcf_model.to_xml_string()
  task = rp.soccer_2v2_boxhead()
  task.initialize_episode_mjcf(np.random.RandomState(0))
  out['tests/golden/pymjcf_soccer_2v2_boxhead_seed0.xml'] = task.root_entity.mjcf_model.to_xml_string()
  task = rp.soccer_2v2_boxhead(randomizer=lambda random_state=None: 0.5)
  task.initialize_episode_mjcf(np.random.RandomState(0))
  out['dm_control_amd/suite/assets/soccer_2v2_boxhead.xml'] = task.root_entity.mjcf_model.to_xml_string()
  return out


if __name__ == '__main__':
  for rel, text in outputs().items():
    with open(os.path.join(ROOT, rel), 'w') as f:
      f.write(text)
    print('wrote', rel, len(text))
