"""Joint randomisation for task initialisation (reference:
dm_control/suite/utils/randomizers.py:35-88), batch-aware."""
import numpy as np

_FREE, _BALL, _SLIDE, _HINGE = 0, 1, 2, 3


def randomize_limited_and_rotational_joints(physics, random=None, env_mask=None):
  """Bounded hinges/sliders ~ U(range); unbounded hinges ~ U(-pi, pi); limited ball joints: a rotation about a
  random axis by U(0, range max); free/ball
  quaternions random unit (the reference draws free-joint quaternions with
  `rand`, ball ones with `randn`; kept); free translations untouched.  With a
  batch, only environments selected by `env_mask` are re-drawn."""
  random = random or np.random
  m = physics.model
  B = physics.batch_size
  qpos = np.asarray(physics.data.qpos).reshape(B, m.nq)
  mask = np.ones(B, dtype=bool) if env_mask is None else np.asarray(env_mask, dtype=bool)
  for e in np.nonzero(mask)[0]:
    for j in range(m.njnt):
      t, a = m.jnt_type[j], m.jnt_qposadr[j]
      lo, hi = m.jnt_range[j]
      if m.jnt_limited[j]:
        if t in (_HINGE, _SLIDE):
          qpos[e, a] = random.uniform(lo, hi)
        elif t == _BALL:
          # random_limited_quaternion (randomizers.py:22-32): axis ~ normalised N(0, I), angle ~ U(0, range max)
          axis = random.randn(3)
          axis /= np.linalg.norm(axis)
          angle = random.rand() * hi
          qpos[e, a] = np.cos(0.5 * angle)
          qpos[e, a + 1:a + 4] = np.sin(0.5 * angle) * axis
      elif t == _HINGE:
        qpos[e, a] = random.uniform(-np.pi, np.pi)
      elif t == _BALL:
        q = random.randn(4)
        qpos[e, a:a + 4] = q / np.linalg.norm(q)
      elif t == _FREE:
        q = random.rand(4)
        qpos[e, a + 3:a + 7] = q / np.linalg.norm(q)
  physics.data.qpos = qpos.reshape(np.shape(physics.data.qpos))
