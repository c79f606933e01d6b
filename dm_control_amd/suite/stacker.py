"""Planar stacker domain (reference: dm_control/suite/stacker.py): stack_2, stack_4.

The manipulator arm with 2 or 4 boxes (box-box, capsule-box and sphere-box contacts, elliptic cones).
Model constants are shared by a batch, so the ghost target's pose (it has no physics) is kept per
environment in the task, as in the manipulator domain."""
import collections
import xml.etree.ElementTree as etree

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import rewards

_CLOSE = .01
_CONTROL_TIMESTEP = .01
_TIME_LIMIT = 10
_ARM_JOINTS = ['arm_root', 'arm_shoulder', 'arm_elbow', 'arm_wrist', 'finger', 'fingertip', 'thumb', 'thumbtip']
_TOUCH_SENSORS = ['palm_touch', 'finger_touch', 'thumb_touch', 'fingertip_touch', 'thumbtip_touch']
TASKS = {}


def make_model(n_boxes):
  """Model XML with only the first n_boxes boxes (stacker.py:41-52)."""
  mjcf = etree.fromstring(common.read_model('stacker.xml'))
  wb = mjcf.find('worldbody')
  for b in range(n_boxes, 4):
    for e in list(wb):
      if e.tag == 'body' and e.get('name') == 'box%d' % b:
        wb.remove(e)
  return etree.tostring(mjcf, encoding='unicode'), None


def _make(n_boxes):
  def factory(fully_observable=True, time_limit=_TIME_LIMIT, random=None, environment_kwargs=None,
              physics_kwargs=None):
    physics = Physics.from_xml_string(*make_model(n_boxes), **common.physics_kwargs('stacker', physics_kwargs))
    task = Stack(n_boxes=n_boxes, fully_observable=fully_observable, random=random)
    return control.Environment(physics, task, control_timestep=_CONTROL_TIMESTEP, time_limit=time_limit,
                               **(environment_kwargs or {}))
  return factory


stack_2, stack_4 = _make(2), _make(4)
TASKS.update(stack_2=(stack_2, None), stack_4=(stack_4, None))


class Physics(physics_lib.Physics):
  target_xz = None   # per-environment ghost target position: (B, 2) or (2,)

  def _q(self, names, field):
    return np.concatenate([getattr(self.named.data, field)[n] for n in names], axis=-1)

  def bounded_joint_pos(self, joint_names):
    joint_pos = self._q(joint_names, 'qpos')
    return np.stack([np.sin(joint_pos), np.cos(joint_pos)], axis=-1)

  def joint_vel(self, joint_names):
    return self._q(joint_names, 'qvel')

  def body_2d_pose(self, body_names, orientation=True):
    """(x, z[, qw, qy]) of one body, or stacked over a list of bodies: (..., n, 2 or 4)."""
    if not isinstance(body_names, str):
      return np.stack([self.body_2d_pose(b, orientation) for b in body_names], axis=-2)
    pos = self.named.data.xpos[body_names][..., [0, 2]]
    if not orientation:
      return pos
    return np.concatenate([pos, self.named.data.xquat[body_names][..., [0, 2]]], axis=-1)

  def target_position(self):
    if self.target_xz is None:
      return self.named.data.xpos['target'][..., [0, 2]]
    return common.array_copy(self.target_xz, dtype=np.float64)      # (a fresh array per observation, like the indexed read above)

  def touch(self):
    return np.log1p(self._q(_TOUCH_SENSORS, 'sensordata'))

  def site_to_target_distance(self, site):
    """Distance from a site to the target site (the target body's origin, at y = body_pos y)."""
    t = self.target_position()
    y = self.named.model.body_pos['target'][1]
    target = np.stack([t[..., 0], np.broadcast_to(y, np.shape(t[..., 0])), t[..., 1]], axis=-1)
    return common.vnorm(self.named.data.site_xpos[site] - target)


class Stack(base.Task):

  def __init__(self, n_boxes, fully_observable, random=None):
    self._n_boxes = n_boxes
    self._box_names = ['box' + str(b) for b in range(n_boxes)]
    self._box_joint_names = ['_'.join([name, dim]) for name in self._box_names for dim in 'xyz']
    self._fully_observable = fully_observable
    super().__init__(random=random)

  def initialize_episode(self, physics):
    uniform, randint = self.random.uniform, self.random.randint
    m = physics.model
    B = physics.batch_size
    jid = lambda n: m.name2id(n, 'joint')
    qadr = lambda n: m.jnt_qposadr[jid(n)]
    arm = [jid(n) for n in _ARM_JOINTS]
    limited = m.jnt_limited[arm].astype(bool)
    lower = np.where(limited, m.jnt_range[arm, 0], -np.pi)
    upper = np.where(limited, m.jnt_range[arm, 1], np.pi)
    box_size = m.geom_size[m.name2id('target', 'geom'), 0]
    target = np.zeros((B, 2))
    todo = np.ones(B, dtype=bool)
    while todo.any():
      qpos = np.array(physics.data.qpos, dtype=np.float64, copy=True).reshape(B, m.nq)
      for e in np.nonzero(todo)[0]:
        qpos[e, [m.jnt_qposadr[j] for j in arm]] = uniform(lower, upper)
        qpos[e, qadr('finger')] = qpos[e, qadr('thumb')]          # symmetric hand
        target_height = 2*randint(self._n_boxes) + 1
        target[e] = (uniform(-.37, .37), box_size * target_height)
        for name in self._box_names:
          qpos[e, qadr(name + '_x')] = uniform(.1, .3)
          qpos[e, qadr(name + '_z')] = uniform(0, .7)
          qpos[e, qadr(name + '_y')] = uniform(0, 2*np.pi)
      physics.data.qpos = qpos.reshape(np.shape(physics.data.qpos))
      with physics.suppress_physics_errors():     # a rejected sample may overflow the contact cap
        physics.after_reset()
      todo &= np.atleast_1d(physics.data.ncon) > 0
    physics.target_xz = target[0] if B == 1 else target
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['arm_pos'] = physics.bounded_joint_pos(_ARM_JOINTS)
    obs['arm_vel'] = physics.joint_vel(_ARM_JOINTS)
    obs['touch'] = physics.touch()
    if self._fully_observable:
      obs['hand_pos'] = physics.body_2d_pose('hand')
      obs['box_pos'] = physics.body_2d_pose(self._box_names)
      obs['box_vel'] = physics.joint_vel(self._box_joint_names)
      obs['target_pos'] = physics.target_position()
    return obs

  def get_reward(self, physics):
    box_size = physics.model.geom_size[physics.model.name2id('target', 'geom'), 0]
    dists = np.stack([physics.site_to_target_distance(name) for name in self._box_names], axis=0)
    box_is_close = rewards.tolerance(dists.min(axis=0), margin=2*box_size)
    hand_is_far = rewards.tolerance(physics.site_to_target_distance('grasp'), bounds=(.1, float('inf')), margin=_CLOSE)
    return box_is_close * hand_is_far
