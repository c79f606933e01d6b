"""Reacher domain (reference: dm_control/suite/reacher.py): easy, hard.

The reference moves the world-fixed `target` geom by rewriting model.geom_pos each
episode (reacher.py:96-102).  Model constants are shared by a whole batch here, so
the per-environment target position / size live in the task; for a static world
geom geom_xpos == geom_pos, so every observable is unchanged."""
import collections

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import randomizers
from dm_control_amd.suite import rewards

_DEFAULT_TIME_LIMIT = 20
_BIG_TARGET = .05
_SMALL_TARGET = .015
TASKS = {}


def get_model_and_assets():
  return common.read_model('reacher.xml'), None


def _make(target_size):
  def factory(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
    physics = Physics.from_xml_string(*get_model_and_assets(), **(physics_kwargs or {}))
    return control.Environment(physics, Reacher(target_size=target_size, random=random), time_limit=time_limit,
                               **(environment_kwargs or {}))
  return factory


easy, hard = _make(_BIG_TARGET), _make(_SMALL_TARGET)
TASKS.update(easy=(easy, 'benchmarking'), hard=(hard, 'benchmarking'))


class Physics(physics_lib.Physics):
  target_xy = None   # (B, 2) or (2,): set by the task at episode start

  def finger_to_target(self):
    """Vector from the finger to the target in the plane (reacher.py:63-66)."""
    finger = self.named.data.geom_xpos['finger'][..., :2]
    target = self.target_xy if self.target_xy is not None else self.named.model.geom_pos['target'][:2]
    return target - finger

  def finger_to_target_dist(self):
    return common.vnorm(self.finger_to_target())


class Reacher(base.Task):

  def __init__(self, target_size, random=None):
    self._target_size = target_size
    super().__init__(random=random)

  def initialize_episode(self, physics):
    randomizers.randomize_limited_and_rotational_joints(physics, self.random)
    B = physics.batch_size
    angle = self.random.uniform(0, 2 * np.pi, B)
    radius = self.random.uniform(.05, .20, B)
    xy = np.stack([radius * np.sin(angle), radius * np.cos(angle)], axis=-1)
    physics.target_xy = xy[0] if B == 1 else xy
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['position'] = physics.position()
    obs['to_target'] = physics.finger_to_target()
    obs['velocity'] = physics.velocity()
    return obs

  def get_reward(self, physics):
    radii = self._target_size + physics.named.model.geom_size['finger'][0]
    return rewards.tolerance(physics.finger_to_target_dist(), (0, radii))
