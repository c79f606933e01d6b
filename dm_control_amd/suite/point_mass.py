"""Point-mass domain (reference: dm_control/suite/point_mass.py): easy, hard.

The two motors act through fixed tendons; `hard` re-draws the tendon coefficients
(model.wrap_prm) each episode so that every control drives a random combination of
the two slide joints.  Model constants are shared by a batch, so with batch_size > 1
all environments of an episode share the drawn gains."""
import collections

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import randomizers
from dm_control_amd.suite import rewards

_DEFAULT_TIME_LIMIT = 20
TASKS = {}


def get_model_and_assets():
  return common.read_model('point_mass.xml'), None


def _make(randomize_gains):
  def factory(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
    physics = Physics.from_xml_string(*get_model_and_assets(), **(physics_kwargs or {}))
    return control.Environment(physics, PointMass(randomize_gains=randomize_gains, random=random),
                               time_limit=time_limit, **(environment_kwargs or {}))
  return factory


easy, hard = _make(False), _make(True)
TASKS.update(easy=(easy, 'benchmarking'), hard=(hard, None))


class Physics(physics_lib.Physics):

  def mass_to_target(self):
    return self.named.data.geom_xpos['target'] - self.named.data.geom_xpos['pointmass']

  def mass_to_target_dist(self):
    return common.vnorm(self.mass_to_target())


class PointMass(base.Task):

  def __init__(self, randomize_gains, random=None):
    self._randomize_gains = randomize_gains
    super().__init__(random=random)

  def initialize_episode(self, physics):
    randomizers.randomize_limited_and_rotational_joints(physics, self.random)
    if self._randomize_gains:
      dir1 = self.random.randn(2)
      dir1 /= np.linalg.norm(dir1)
      parallel = True
      while parallel:   # a second actuation direction that is not too parallel to the first
        dir2 = self.random.randn(2)
        dir2 /= np.linalg.norm(dir2)
        parallel = abs(np.dot(dir1, dir2)) > 0.9
      physics.model.wrap_prm[[0, 1]] = dir1
      physics.model.wrap_prm[[2, 3]] = dir2
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['position'] = physics.position()
    obs['velocity'] = physics.velocity()
    return obs

  def get_reward(self, physics):
    target_size = physics.named.model.geom_size['target'][0]
    near_target = rewards.tolerance(physics.mass_to_target_dist(), bounds=(0, target_size), margin=target_size)
    control_reward = rewards.tolerance(physics.control(), margin=1, value_at_margin=0,
                                       sigmoid='quadratic').mean(axis=-1)
    small_control = (control_reward + 4) / 5
    return near_target * small_control
