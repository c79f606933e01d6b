"""The numpy mirror of the device joint randomiser (tests/philox_mirror.py): Philox4x32-10 on the published
known-answer vectors, and the draw rules of suite/utils/randomizers.py:35-88 as distributions."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import philox_mirror as pm  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_philox4x32_10_known_answers():
  # Random123 kat_vectors, "philox4x32 10" rows
  F = 0xFFFFFFFF
  assert pm.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
  assert pm.philox4x32_10((F, F, F, F), (F, F)) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
  assert pm.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
      (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_draw_rules_follow_the_reference_randomizer():
  from dm_control_amd import mjcf_compiler as mc
  from dm_control_amd.suite import common
  m = mc.compile_xml(common.read_model('humanoid.xml'))
  B = 400
  q = np.tile(m.qpos0, (B, 1))
  draw = np.zeros(B, int)
  pm.randomize_joints(m, q, seed=7, draw=draw)
  assert (draw == 1).all()
  np.testing.assert_array_equal(q[:, :3], np.tile(m.qpos0[:3], (B, 1)))      # free translation untouched
  np.testing.assert_allclose(np.linalg.norm(q[:, 3:7], axis=1), 1, atol=1e-12)
  assert (q[:, 3:7] > 0).all()                                                # normalised U(0,1)^4 (randomizers.py:84-88)
  for j in range(1, m.njnt):
    a = m.jnt_qposadr[j]
    lo, hi = m.jnt_range[j]
    assert m.jnt_limited[j]
    assert (q[:, a] >= lo).all() and (q[:, a] <= hi).all()
    # uniform in the range: mean and spread within 5 sigma of U(lo, hi)
    assert abs(q[:, a].mean() - 0.5 * (lo + hi)) < 5 * (hi - lo) / np.sqrt(12 * B)
    assert abs(q[:, a].std() - (hi - lo) / np.sqrt(12)) < 0.15 * (hi - lo) / np.sqrt(12)
  # a second draw of a subset leaves the rest alone and never repeats a value
  q2 = q.copy()
  mask = np.arange(B) % 2 == 0
  pm.randomize_joints(m, q2, seed=7, draw=draw, env_mask=mask)
  np.testing.assert_array_equal(q2[~mask], q[~mask])
  assert (q2[mask, 7:] != q[mask, 7:]).all()
  assert (draw == 1 + mask).all()
  # only the limited joints (suite/cheetah.py:66-69)
  c = mc.compile_xml(common.read_model('cheetah.xml'))
  qc = np.tile(c.qpos0, (8, 1))
  pm.randomize_joints(c, qc, seed=1, draw=np.zeros(8, int), flags=pm.LIMITED)
  np.testing.assert_array_equal(qc[:, :3], 0)
  assert (qc[:, 3:] != 0).all()
  # unlimited hinges in [-pi, pi], sphere-uniform free quaternion on request (suite/quadruped.py:243-246)
  qd = mc.compile_xml(common.read_model('quadruped.xml')) if os.path.exists(os.path.join(ROOT, 'dm_control_amd/suite/assets/quadruped.xml')) else None
  if qd is not None:
    qq = np.tile(qd.qpos0, (300, 1))
    pm.randomize_joints(qd, qq, seed=3, draw=np.zeros(300, int), flags=pm.QUATERNION | pm.FREE_NORMAL)
    assert (qq[:, 3:7] < 0).any() and abs(qq[:, 3:7].mean()) < 0.1


def test_consecutive_seeds_are_independent_streams():
  """Seeds 0, 1, 2 ... must not be permutations of one another over a power-of-two batch (keying by seed ^ env was)."""
  from dm_control_amd import mjcf_compiler as mc
  from dm_control_amd.suite import common
  c = mc.compile_xml(common.read_model('cheetah.xml'))
  B = 16
  sets = []
  for seed in (0, 1, 2, 5):
    q = np.tile(c.qpos0, (B, 1))
    pm.randomize_joints(c, q, seed=seed, draw=np.zeros(B, int), flags=pm.LIMITED)
    sets.append({tuple(np.round(r, 12)) for r in q})
  for i in range(len(sets)):
    for j in range(i + 1, len(sets)):
      assert not sets[i] & sets[j]
