"""Humanoid_CMU domain (reference: dm_control/suite/humanoid_CMU.py): stand, walk, run.

62 degrees of freedom, 56 motors, capsule limbs, sphere toes and two ellipsoid hands
that collide with the whole body (1118 candidate pairs): the ellipsoid pairs use the
iterative support-function narrow phase (csrc/step_core.h `ellipsoid_pair`)."""
import collections

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import randomizers
from dm_control_amd.suite import rewards

_DEFAULT_TIME_LIMIT = 20
_CONTROL_TIMESTEP = 0.02
_STAND_HEIGHT = 1.4
_WALK_SPEED = 1
_RUN_SPEED = 10
TASKS = {}


def get_model_and_assets():
  return common.read_model('humanoid_CMU.xml'), None


def _make(move_speed):
  def factory(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
    physics = Physics.from_xml_string(*get_model_and_assets(), **common.physics_kwargs('humanoid_CMU', physics_kwargs))
    task = HumanoidCMU(move_speed=move_speed, random=random)
    return control.Environment(physics, task, time_limit=time_limit, control_timestep=_CONTROL_TIMESTEP,
                               **(environment_kwargs or {}))
  return factory


stand = _make(0)
walk = _make(_WALK_SPEED)
run = _make(_RUN_SPEED)
TASKS.update(stand=(stand, None), walk=(walk, None), run=(run, None))


class Physics(physics_lib.Physics):

  def thorax_upright(self):
    """Projection of the thorax y axis on the world z axis."""
    return self.named.data.xmat['thorax', 'zy']

  def head_height(self):
    return self.named.data.xpos['head', 'z']

  def center_of_mass_position(self):
    return self.named.data.subtree_com['thorax'].copy()

  def center_of_mass_velocity(self):
    return self.named.data.sensordata['thorax_subtreelinvel'].copy()

  def torso_vertical_orientation(self):
    return self.named.data.xmat['thorax', ['zx', 'zy', 'zz']]

  def joint_angles(self):
    return self.data.qpos[..., 7:].copy()

  def extremities(self):
    """Hand / foot positions in the egocentric thorax frame."""
    xmat = self.named.data.xmat['thorax']
    frame = xmat.reshape(xmat.shape[:-1] + (3, 3))
    torso_pos = self.named.data.xpos['thorax']
    out = []
    for side in ('l', 'r'):
      for limb in ('hand', 'foot'):
        d = self.named.data.xpos[side + limb] - torso_pos
        out.append(np.einsum('...i,...ij->...j', d, frame))
    return np.concatenate(out, axis=-1)


class HumanoidCMU(base.Task):

  def __init__(self, move_speed, random=None):
    self._move_speed = move_speed
    super().__init__(random=random)

  def initialize_episode(self, physics):
    # rejection-sample a collision-free configuration (humanoid_CMU.py:137-145); for a
    # batch only the still-penetrating environments are re-drawn
    todo = np.ones(physics.batch_size, dtype=bool)
    while todo.any():
      randomizers.randomize_limited_and_rotational_joints(physics, self.random, env_mask=todo)
      with physics.suppress_physics_errors():   # a rejected sample may overflow the contact cap
        physics.after_reset()
      todo &= np.atleast_1d(physics.data.ncon) > 0
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['joint_angles'] = physics.joint_angles()
    obs['head_height'] = physics.head_height()
    obs['extremities'] = physics.extremities()
    obs['torso_vertical'] = physics.torso_vertical_orientation()
    obs['com_velocity'] = physics.center_of_mass_velocity()
    obs['velocity'] = physics.velocity()
    return obs

  def get_reward(self, physics):
    standing = rewards.tolerance(physics.head_height(), bounds=(_STAND_HEIGHT, float('inf')),
                                 margin=_STAND_HEIGHT / 4)
    upright = rewards.tolerance(physics.thorax_upright(), bounds=(0.9, float('inf')), sigmoid='linear',
                                margin=1.9, value_at_margin=0)
    stand_reward = standing * upright
    small_control = rewards.tolerance(physics.control(), margin=1, value_at_margin=0,
                                      sigmoid='quadratic').mean(axis=-1)
    small_control = (4 + small_control) / 5
    horizontal = physics.center_of_mass_velocity()[..., [0, 1]]
    if self._move_speed == 0:
      dont_move = rewards.tolerance(horizontal, margin=2).mean(axis=-1)
      return small_control * stand_reward * dont_move
    speed = np.linalg.norm(horizontal, axis=-1)
    move = rewards.tolerance(speed, bounds=(self._move_speed, float('inf')), margin=self._move_speed,
                             value_at_margin=0, sigmoid='linear')
    return small_control * stand_reward * (5 * move + 1) / 6
