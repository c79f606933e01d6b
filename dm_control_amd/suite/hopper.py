"""Hopper domain (reference: dm_control/suite/hopper.py): stand, hop."""
import collections

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import randomizers
from dm_control_amd.suite import rewards

_CONTROL_TIMESTEP = .02
_DEFAULT_TIME_LIMIT = 20
_STAND_HEIGHT = 0.6
_HOP_SPEED = 2
TASKS = {}


def get_model_and_assets():
  return common.read_model('hopper.xml'), None


def _make(hopping):
  def factory(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
    physics = Physics.from_xml_string(*get_model_and_assets(), **(physics_kwargs or {}))
    return control.Environment(physics, Hopper(hopping=hopping, random=random), time_limit=time_limit,
                               control_timestep=_CONTROL_TIMESTEP, **(environment_kwargs or {}))
  return factory


stand, hop = _make(False), _make(True)
TASKS.update(stand=(stand, 'benchmarking'), hop=(hop, 'benchmarking'))


class Physics(physics_lib.Physics):

  def height(self):
    """Torso height above the foot (inertial frames)."""
    return self.named.data.xipos['torso', 'z'] - self.named.data.xipos['foot', 'z']

  def speed(self):
    return self.named.data.sensordata['torso_subtreelinvel'][..., 0]

  def touch(self):
    return np.log1p(self.named.data.sensordata[['touch_toe', 'touch_heel']])


class Hopper(base.Task):

  def __init__(self, hopping, random=None):
    self._hopping = hopping
    super().__init__(random=random)

  def initialize_episode(self, physics):
    randomizers.randomize_limited_and_rotational_joints(physics, self.random)
    self._timeout_progress = 0
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['position'] = physics.data.qpos[..., 1:].copy()
    obs['velocity'] = physics.velocity()
    obs['touch'] = physics.touch()
    return obs

  def get_reward(self, physics):
    standing = rewards.tolerance(physics.height(), (_STAND_HEIGHT, 2))
    if self._hopping:
      hopping = rewards.tolerance(physics.speed(), bounds=(_HOP_SPEED, float('inf')), margin=_HOP_SPEED / 2,
                                  value_at_margin=0.5, sigmoid='linear')
      return standing * hopping
    small_control = rewards.tolerance(physics.control(), margin=1, value_at_margin=0,
                                      sigmoid='quadratic').mean(axis=-1)
    return standing * (small_control + 4) / 5
