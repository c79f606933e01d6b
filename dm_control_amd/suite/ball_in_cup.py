"""Ball-in-cup domain (reference: dm_control/suite/ball_in_cup.py): catch.

The ball hangs from the cup on a string: a site-to-site spatial tendon with a length
limit (a constraint row when the string is taut)."""
import collections

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common

_DEFAULT_TIME_LIMIT = 20
_CONTROL_TIMESTEP = .02
TASKS = {}


def get_model_and_assets():
  return common.read_model('ball_in_cup.xml'), None


def catch(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
  physics = Physics.from_xml_string(*get_model_and_assets(), **(physics_kwargs or {}))
  return control.Environment(physics, BallInCup(random=random), time_limit=time_limit,
                             control_timestep=_CONTROL_TIMESTEP, **(environment_kwargs or {}))


TASKS.update(catch=(catch, 'benchmarking'))


class Physics(physics_lib.Physics):

  def ball_to_target(self):
    """(x, z) vector from the ball to the target site."""
    target = self.named.data.site_xpos['target'][..., [0, 2]]
    ball = self.named.data.xpos['ball'][..., [0, 2]]
    return target - ball

  def in_target(self):
    """1 if the ball is inside the target box (ball_in_cup.py:67-72)."""
    d = np.abs(self.ball_to_target())
    target_size = self.named.model.site_size['target'][[0, 2]]
    ball_size = self.named.model.geom_size['ball'][0]
    return common.asarray(np.all(d < target_size - ball_size, axis=-1), dtype=np.float64)


class BallInCup(base.Task):

  def initialize_episode(self, physics):
    # rejection-sample a collision-free ball position; with a batch only the still
    # penetrating environments are re-drawn
    m = physics.model
    ax = m.jnt_qposadr[m.name2id('ball_x', 'joint')]
    az = m.jnt_qposadr[m.name2id('ball_z', 'joint')]
    B = physics.batch_size
    todo = np.ones(B, dtype=bool)
    while todo.any():
      qpos = np.array(physics.data.qpos, dtype=np.float64, copy=True).reshape(B, m.nq)
      for e in np.nonzero(todo)[0]:
        qpos[e, ax] = self.random.uniform(-.2, .2)
        qpos[e, az] = self.random.uniform(.2, .5)
      physics.data.qpos = qpos.reshape(np.shape(physics.data.qpos))
      with physics.suppress_physics_errors():   # a rejected sample may overflow the contact cap
        physics.after_reset()
      todo &= np.atleast_1d(physics.data.ncon) > 0
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['position'] = physics.position()
    obs['velocity'] = physics.velocity()
    return obs

  def get_reward(self, physics):
    return physics.in_target()
