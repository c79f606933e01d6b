"""Loader for tests/golden/cmu2019_mocap.json (MuJoCo-generated kinematics of the CMU 2019 walker held by the
reference: locomotion/mocap/test_00{1,2}.textproto; extracted by scripts/make_mocap_golden.py)."""
import json
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cmu2019_mocap.json')


def load():
  with open(PATH) as f:
    g = json.load(f)
  frames = [fr for clip in g['clips'] for fr in clip['frames']]
  out = {k: np.array([fr[k] for fr in frames]) for k in frames[0]}
  out['joint_order'] = g['joint_order']
  out['end_effector_bodies'] = g['end_effector_bodies']
  out['appendage_bodies'] = g['appendage_bodies']
  return out


def qpos_of_frames(model, g, root_joint, prefix=''):
  """(nframes, nq): root free joint <- position / quaternion, hinge joints by name <- joints
  (reference_pose/utils.py:103-117 `set_walker` via walker.set_pose + bind(mocap_joints).qpos); `prefix`: PyMJCF's
  name scope of the walker in a composed model ('walker/')."""
  n = g['position'].shape[0]
  q = np.tile(np.asarray(model.qpos0, dtype=np.float64), (n, 1))
  adr = int(model.jnt_qposadr[model.name2id(root_joint, 'joint')])
  q[:, adr:adr + 3] = g['position']
  q[:, adr + 3:adr + 7] = g['quaternion']
  for k, name in enumerate(g['joint_order']):
    q[:, int(model.jnt_qposadr[model.name2id(prefix + name, 'joint')])] = g['joints'][:, k]
  return q


def tracking_bodies(model, root_body='root'):
  """mocap_tracking_bodies (walkers/cmu_humanoid.py:331-336): every walker body but `root`, document order
  (= body id order: MuJoCo numbers bodies depth first in document order)."""
  r = model.name2id(root_body, 'body')
  return list(range(r + 1, model.nbody))


def egocentric(xpos, xmat, root, bodies):
  """cmu_humanoid.py:473-482: (xpos[b] - xpos[root]) . xmat[root], flattened."""
  R = np.asarray(xmat)[9 * root:9 * root + 9].reshape(3, 3)
  x = np.asarray(xpos).reshape(-1, 3)
  return np.concatenate([(x[b] - x[root]) @ R for b in bodies])
