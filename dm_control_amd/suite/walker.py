"""Planar walker domain (reference: dm_control/suite/walker.py): stand, walk, run."""
import collections

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import randomizers
from dm_control_amd.suite import rewards

_DEFAULT_TIME_LIMIT = 25
_CONTROL_TIMESTEP = .025
_STAND_HEIGHT = 1.2     # torso height above which the stand reward is 1
_WALK_SPEED = 1
_RUN_SPEED = 8
TASKS = {}


def get_model_and_assets():
  return common.read_model('walker.xml'), None


def _make(move_speed):
  def factory(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
    physics = Physics.from_xml_string(*get_model_and_assets(), **(physics_kwargs or {}))
    return control.Environment(physics, PlanarWalker(move_speed=move_speed, random=random), time_limit=time_limit,
                               control_timestep=_CONTROL_TIMESTEP, **(environment_kwargs or {}))
  return factory


stand, walk, run = _make(0), _make(_WALK_SPEED), _make(_RUN_SPEED)
TASKS.update(stand=(stand, 'benchmarking'), walk=(walk, 'benchmarking'), run=(run, 'benchmarking'))


class Physics(physics_lib.Physics):

  def torso_upright(self):
    return self.named.data.xmat['torso', 'zz']

  def torso_height(self):
    return self.named.data.xpos['torso', 'z']

  def horizontal_velocity(self):
    return self.named.data.sensordata['torso_subtreelinvel'][..., 0]

  def orientations(self):
    o = self.named.data.xmat[1:, ['xx', 'xz']]
    return o.reshape(o.shape[:-2] + (-1,))


class PlanarWalker(base.Task):

  def __init__(self, move_speed, random=None):
    self._move_speed = move_speed
    super().__init__(random=random)

  def initialize_episode(self, physics):
    randomizers.randomize_limited_and_rotational_joints(physics, self.random)
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['orientations'] = physics.orientations()
    obs['height'] = physics.torso_height()
    obs['velocity'] = physics.velocity()
    return obs

  def get_reward(self, physics):
    standing = rewards.tolerance(physics.torso_height(), bounds=(_STAND_HEIGHT, float('inf')),
                                 margin=_STAND_HEIGHT / 2)
    upright = (1 + physics.torso_upright()) / 2
    stand_reward = (3 * standing + upright) / 4
    if self._move_speed == 0:
      return stand_reward
    move = rewards.tolerance(physics.horizontal_velocity(), bounds=(self._move_speed, float('inf')),
                             margin=self._move_speed / 2, value_at_margin=0.5, sigmoid='linear')
    return stand_reward * (5 * move + 1) / 6
