"""Finger domain (reference: dm_control/suite/finger.py): spin, turn_easy, turn_hard.

Exercises elliptic friction cones, dof friction loss, framepos sensors and touch
sensors on ellipsoid sites.  The reference moves the world-fixed `target` site by
rewriting model.site_pos per episode (finger.py:163-170); with one set of model
constants per batch the per-environment target lives in the task instead (for a
world site site_xpos == site_pos, so the `target` observations are unchanged)."""
import collections

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import randomizers

_DEFAULT_TIME_LIMIT = 20
_CONTROL_TIMESTEP = .02
_EASY_TARGET_SIZE = 0.07
_HARD_TARGET_SIZE = 0.03
_SPIN_VELOCITY = 15.0   # spinning faster than this (rad/s) counts as spinning
TASKS = {}


def get_model_and_assets():
  return common.read_model('finger.xml'), None


def _make(task_factory):
  def factory(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
    physics = Physics.from_xml_string(*get_model_and_assets(), **(physics_kwargs or {}))
    return control.Environment(physics, task_factory(random), time_limit=time_limit,
                               control_timestep=_CONTROL_TIMESTEP, **(environment_kwargs or {}))
  return factory


class Physics(physics_lib.Physics):
  target_xz = None       # per-environment target position, (B, 2) or (2,); None: the model's
  target_radius = None

  def _sensor(self, name):
    return self.named.data.sensordata[name]

  def touch(self):
    """Log-scaled signals of the two touch sensors (finger.py:90-92)."""
    return np.log1p(self.named.data.sensordata[['touchtop', 'touchbottom']])

  def hinge_velocity(self):
    return self._sensor('hinge_velocity')[..., 0]

  def tip_position(self):
    """(x, z) of the tip relative to the hinge."""
    return self._sensor('tip')[..., [0, 2]] - self._sensor('spinner')[..., [0, 2]]

  def bounded_position(self):
    return np.concatenate([self._sensor('proximal'), self._sensor('distal'), self.tip_position()], axis=-1)

  def velocity(self):
    return np.concatenate([self._sensor('proximal_velocity'), self._sensor('distal_velocity'),
                           self._sensor('hinge_velocity')], axis=-1)

  def target_position(self):
    target = self.target_xz if self.target_xz is not None else self._sensor('target')[..., [0, 2]]
    return target - self._sensor('spinner')[..., [0, 2]]

  def to_target(self):
    return self.target_position() - self.tip_position()

  def dist_to_target(self):
    """Signed distance to the target surface, negative inside (finger.py:122-125)."""
    radius = self.target_radius if self.target_radius is not None else self.named.model.site_size['target'][0]
    return common.vnorm(self.to_target()) - radius


def _set_random_joint_angles(physics, random, max_attempts=1000):
  """Random collision-free joint configuration; with a batch only the still
  colliding environments are re-drawn (finger.py:218-230)."""
  todo = np.ones(physics.batch_size, dtype=bool)
  for _ in range(max_attempts):
    randomizers.randomize_limited_and_rotational_joints(physics, random, env_mask=todo)
    with physics.suppress_physics_errors():   # a rejected sample may overflow the contact cap
      physics.after_reset()
    todo &= np.atleast_1d(physics.data.ncon) > 0
    if not todo.any():
      return
  raise RuntimeError('Could not find a collision-free state after {} attempts'.format(max_attempts))


class Spin(base.Task):

  def initialize_episode(self, physics):
    physics.named.model.dof_damping['hinge'] = .03
    _set_random_joint_angles(physics, self.random)
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['position'] = physics.bounded_position()
    obs['velocity'] = physics.velocity()
    obs['touch'] = physics.touch()
    return obs

  def get_reward(self, physics):
    return common.asarray(physics.hinge_velocity() <= -_SPIN_VELOCITY, dtype=np.float64)


class Turn(base.Task):

  def __init__(self, target_radius, random=None):
    self._target_radius = target_radius
    super().__init__(random=random)

  def initialize_episode(self, physics):
    B = physics.batch_size
    target_angle = self.random.uniform(-np.pi, np.pi, B)
    # hinge anchor: the spinner body's origin (its joint sits at the body origin)
    anchor = np.asarray(physics.named.data.xpos['spinner']).reshape(B, 3)
    radius = physics.named.model.geom_size['cap1'].sum()
    xz = np.stack([anchor[:, 0] + radius * np.sin(target_angle), anchor[:, 2] + radius * np.cos(target_angle)], axis=-1)
    physics.target_xz = xz[0] if B == 1 else xz
    physics.target_radius = self._target_radius
    _set_random_joint_angles(physics, self.random)
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['position'] = physics.bounded_position()
    obs['velocity'] = physics.velocity()
    obs['touch'] = physics.touch()
    obs['target_position'] = physics.target_position()
    obs['dist_to_target'] = physics.dist_to_target()
    return obs

  def get_reward(self, physics):
    return common.asarray(physics.dist_to_target() <= 0, dtype=np.float64)


spin = _make(lambda random: Spin(random=random))
turn_easy = _make(lambda random: Turn(target_radius=_EASY_TARGET_SIZE, random=random))
turn_hard = _make(lambda random: Turn(target_radius=_HARD_TARGET_SIZE, random=random))
TASKS.update(spin=(spin, 'benchmarking'), turn_easy=(turn_easy, 'benchmarking'), turn_hard=(turn_hard, 'benchmarking'))
