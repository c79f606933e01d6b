"""Humanoid domain (reference: dm_control/suite/humanoid.py): stand, walk, run,
run_pure_state."""
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
_STAND_HEIGHT = 1.4
_WALK_SPEED = 1
_RUN_SPEED = 10
TASKS = {}


def get_model_and_assets():
  return common.read_model('humanoid.xml'), None


def _make(move_speed, pure_state):
  def factory(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
    physics = Physics.from_xml_string(*get_model_and_assets(), **common.physics_kwargs('humanoid', physics_kwargs))
    task = Humanoid(move_speed=move_speed, pure_state=pure_state, random=random)
    return control.Environment(physics, task, time_limit=time_limit, control_timestep=_CONTROL_TIMESTEP,
                               **(environment_kwargs or {}))
  return factory


stand = _make(0, False)
walk = _make(_WALK_SPEED, False)
run = _make(_RUN_SPEED, False)
run_pure_state = _make(_RUN_SPEED, True)
TASKS.update(stand=(stand, 'benchmarking'), walk=(walk, 'benchmarking'), run=(run, 'benchmarking'),
             run_pure_state=(run_pure_state, None))


class Physics(physics_lib.Physics):
  # body / sensor names of the model; the CMU humanoid (suite/humanoid_CMU.py) reuses everything below
  # with its own names
  TORSO = 'torso'
  UPRIGHT_AXIS = 'zz'                      # torso axis whose world-z component measures uprightness
  SIDES = ('left_', 'right_')
  COM_VELOCITY_SENSOR = 'torso_subtreelinvel'

  def torso_upright(self):
    return self.named.data.xmat[self.TORSO, self.UPRIGHT_AXIS]

  def head_height(self):
    return self.named.data.xpos['head', 'z']

  def center_of_mass_position(self):
    return self.named.data.subtree_com[self.TORSO].copy()

  def center_of_mass_velocity(self):
    return self.named.data.sensordata[self.COM_VELOCITY_SENSOR].copy()

  def torso_vertical_orientation(self):
    return self.named.data.xmat[self.TORSO, ['zx', 'zy', 'zz']]

  def joint_angles(self):
    return self.data.qpos[..., 7:].copy()

  def extremities(self):
    """Hand / foot positions in the egocentric torso frame."""
    xmat = self.named.data.xmat[self.TORSO]
    frame = xmat.reshape(xmat.shape[:-1] + (3, 3))
    torso_pos = self.named.data.xpos[self.TORSO]
    out = []
    for side in self.SIDES:
      for limb in ('hand', 'foot'):
        d = self.named.data.xpos[side + limb] - torso_pos
        out.append(common.vecmat(d, frame))
    return np.concatenate(out, axis=-1)


class Humanoid(base.Task):

  def __init__(self, move_speed, pure_state, random=None):
    self._move_speed = move_speed
    self._pure_state = pure_state
    super().__init__(random=random)

  def initialize_episode(self, physics):
    # rejection-sample a collision-free configuration (humanoid.py:160-165); for a
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
    if self._pure_state:
      obs['position'] = physics.position()
      obs['velocity'] = physics.velocity()
    else:
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
    upright = rewards.tolerance(physics.torso_upright(), bounds=(0.9, float('inf')), sigmoid='linear',
                                margin=1.9, value_at_margin=0)
    stand_reward = standing * upright
    small_control = rewards.tolerance(physics.control(), margin=1, value_at_margin=0,
                                      sigmoid='quadratic').mean(axis=-1)
    small_control = (4 + small_control) / 5
    horizontal = physics.center_of_mass_velocity()[..., [0, 1]]
    if self._move_speed == 0:
      dont_move = rewards.tolerance(horizontal, margin=2).mean(axis=-1)
      return small_control * stand_reward * dont_move
    speed = common.vnorm(horizontal)
    move = rewards.tolerance(speed, bounds=(self._move_speed, float('inf')), margin=self._move_speed,
                             value_at_margin=0, sigmoid='linear')
    move = (5 * move + 1) / 6
    return small_control * stand_reward * move
