"""Base class of the suite tasks (reference: dm_control/suite/base.py:24-92),
batch-aware: `action` may be (nu,) or (B, nu)."""
import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control


class Task(control.Task):

  def __init__(self, random=None):
    if not isinstance(random, np.random.RandomState):
      random = np.random.RandomState(random)
    self._random = random
    self._visualize_reward = False

  @property
  def random(self):
    return self._random

  def action_spec(self, physics):
    return physics_lib.action_spec(physics)

  def initialize_episode(self, physics):
    self.after_step(physics)

  def before_step(self, action, physics):
    action = getattr(action, 'continuous_actions', action)
    physics.set_control(action)

  def after_step(self, physics):
    pass  # reward colouring is rendering-only (out of scope)

  def observation_spec(self, physics):
    raise NotImplementedError()

  @property
  def visualize_reward(self):
    return self._visualize_reward

  @visualize_reward.setter
  def visualize_reward(self, value):
    if not isinstance(value, bool):
      raise ValueError('Expected a boolean, got {}.'.format(type(value)))
    self._visualize_reward = value
