"""Planar manipulator domain (reference: dm_control/suite/manipulator.py): bring_ball, bring_peg,
insert_ball, insert_peg.

Elliptic cones, a fixed-tendon transmission (grasp), a tendon equality constraint (finger / thumb
coupling), box touch sites, capsule-box contacts between the peg / arm and the slot.  Model constants are shared by a batch: the ghost target's pose is kept per environment
in the task (it has no physics); the receptacle of `insert_ball` collides, so its pose is drawn once
per episode for the whole batch."""
import collections
import xml.etree.ElementTree as etree

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import rewards

_CLOSE = .01    # (metres) distance below which a thing counts as close to another
_CONTROL_TIMESTEP = .01
_TIME_LIMIT = 10
_P_IN_HAND = .1
_P_IN_TARGET = .1
_ARM_JOINTS = ['arm_root', 'arm_shoulder', 'arm_elbow', 'arm_wrist', 'finger', 'fingertip', 'thumb', 'thumbtip']
_ALL_PROPS = frozenset(['ball', 'target_ball', 'cup', 'peg', 'target_peg', 'slot'])
_TOUCH_SENSORS = ['palm_touch', 'finger_touch', 'thumb_touch', 'fingertip_touch', 'thumbtip_touch']
TASKS = {}


def make_model(use_peg, insert):
  """Model XML with only the props the task needs (manipulator.py:47-69)."""
  mjcf = etree.fromstring(common.read_model('manipulator.xml'))
  if use_peg:
    required = ['peg', 'target_peg'] + (['slot'] if insert else [])
  else:
    required = ['ball', 'target_ball'] + (['cup'] if insert else [])
  parents = {c: p for p in mjcf.iter() for c in p}
  for unused in _ALL_PROPS.difference(required):
    for e in mjcf.iter('body'):
      if e.get('name') == unused:
        parents[e].remove(e)
        break
  return etree.tostring(mjcf, encoding='unicode'), None


def _make(use_peg, insert):
  def factory(fully_observable=True, time_limit=_TIME_LIMIT, random=None, environment_kwargs=None,
              physics_kwargs=None):
    kw = dict(nconmax=64)     # a folded arm touches itself in many places (40 still overflowed 1-6 times per 77 k env-steps)
    kw.update(physics_kwargs or {})
    physics = Physics.from_xml_string(*make_model(use_peg, insert), **kw)
    task = Bring(use_peg=use_peg, insert=insert, fully_observable=fully_observable, random=random)
    return control.Environment(physics, task, control_timestep=_CONTROL_TIMESTEP, time_limit=time_limit,
                               **(environment_kwargs or {}))
  return factory


bring_ball, bring_peg, insert_ball, insert_peg = _make(False, False), _make(True, False), _make(False, True), _make(True, True)
TASKS.update(bring_ball=(bring_ball, 'benchmarking'), bring_peg=(bring_peg, None), insert_ball=(insert_ball, None),
             insert_peg=(insert_peg, None))


def _quat_y(angle):
  """(qw, qy) of a rotation by `angle` about the y axis."""
  return np.stack([np.cos(angle / 2), np.sin(angle / 2)], axis=-1)


class Physics(physics_lib.Physics):
  target_pose = None   # per-environment ghost target: (B, 3) or (3,) = x, z, angle about y

  def _q(self, names, field):
    return np.concatenate([getattr(self.named.data, field)[n] for n in names], axis=-1)

  def bounded_joint_pos(self, joint_names):
    """(sin, cos) of each joint angle: (..., n, 2)."""
    joint_pos = self._q(joint_names, 'qpos')
    return np.stack([np.sin(joint_pos), np.cos(joint_pos)], axis=-1)

  def joint_vel(self, joint_names):
    return self._q(joint_names, 'qvel')

  def body_2d_pose(self, body_name, orientation=True):
    """(x, z[, qw, qy]) of a body."""
    pos = self.named.data.xpos[body_name][..., [0, 2]]
    if not orientation:
      return pos
    ori = self.named.data.xquat[body_name][..., [0, 2]]
    return np.concatenate([pos, ori], axis=-1)

  def target_2d_pose(self, target_body):
    if self.target_pose is None:
      return self.body_2d_pose(target_body)
    tp = common.asarray(self.target_pose)
    return np.concatenate([tp[..., :2], _quat_y(tp[..., 2])], axis=-1)

  def touch(self):
    return np.log1p(self._q(_TOUCH_SENSORS, 'sensordata'))

  def site_distance(self, site1, site2):
    d = self.named.data.site_xpos[site1] - self.named.data.site_xpos[site2]
    return common.vnorm(d)

  def target_site_distance(self, site, target_body, offset_local):
    """Distance from `site` to a point rigidly attached to the per-environment ghost target
    (offset_local: the target site's position in the target body frame)."""
    tp = common.asarray(self.target_pose)
    x, z, ang = tp[..., 0], tp[..., 1], tp[..., 2]
    c, s = np.cos(ang), np.sin(ang)
    ox, oy, oz = offset_local
    # rotation by `ang` about y: (x, z) -> (c x + s z, -s x + c z)
    px = x + c * ox + s * oz
    pz = z - s * ox + c * oz
    sp = self.named.data.site_xpos[site]
    py = self.named.model.body_pos[target_body][1] + oy
    target = np.stack([px, np.broadcast_to(py, np.shape(px)), pz], axis=-1)
    return common.vnorm(sp - target)


class Bring(base.Task):

  def __init__(self, use_peg, insert, fully_observable, random=None):
    self._use_peg = use_peg
    self._target = 'target_peg' if use_peg else 'target_ball'
    self._object = 'peg' if use_peg else 'ball'
    self._object_joints = ['_'.join([self._object, dim]) for dim in 'xzy']
    self._receptacle = 'slot' if use_peg else 'cup'
    self._insert = insert
    self._fully_observable = fully_observable
    super().__init__(random=random)

  def initialize_episode(self, physics):
    uniform, choice = self.random.uniform, self.random.choice
    m = physics.model
    B = physics.batch_size
    jid = lambda n: m.name2id(n, 'joint')
    qadr = lambda n: m.jnt_qposadr[jid(n)]
    arm = [jid(n) for n in _ARM_JOINTS]
    limited = m.jnt_limited[arm].astype(bool)
    lower = np.where(limited, m.jnt_range[arm, 0], -np.pi)
    upper = np.where(limited, m.jnt_range[arm, 1], np.pi)
    # The receptacle collides: its pose is a model constant, one per episode for the whole batch.  A single environment
    # draws it where the reference does (manipulator.py:207-216: inside the rejection loop, after the arm angles), so
    # that the same seed gives the same episode; a batch draws it once, ahead of the loop.
    shared_target = None
    rb = m.name2id(self._receptacle, 'body') if self._insert else None

    def place_receptacle(pose):
      m.body_pos[rb, [0, 2]] = pose[:2]
      m.body_quat[rb] = [np.cos(pose[2] / 2), 0, np.sin(pose[2] / 2), 0]
    if self._insert and B > 1:
      shared_target = (uniform(-.4, .4), uniform(.1, .4), uniform(-np.pi / 3, np.pi / 3))
      place_receptacle(shared_target)
    target = np.zeros((B, 3))
    todo = np.ones(B, dtype=bool)
    while todo.any():
      idx = np.nonzero(todo)[0]
      qpos = np.array(physics.data.qpos, dtype=np.float64, copy=True).reshape(B, m.nq)
      qvel = np.array(physics.data.qvel, dtype=np.float64, copy=True).reshape(B, m.nv)
      kinds = {}
      for e in idx:
        qpos[e, [m.jnt_qposadr[j] for j in arm]] = uniform(lower, upper)
        qpos[e, qadr('finger')] = qpos[e, qadr('thumb')]          # symmetric hand
        if shared_target is not None:
          target[e] = shared_target
        elif self._insert:
          target[e] = (uniform(-.4, .4), uniform(.1, .4), uniform(-np.pi / 3, np.pi / 3))
          place_receptacle(target[e])
        else:
          target[e] = (uniform(-.4, .4), uniform(.1, .4), uniform(-np.pi, np.pi))
        kinds[e] = choice(['in_hand', 'in_target', 'uniform'], p=[_P_IN_HAND, _P_IN_TARGET, 1 - _P_IN_HAND - _P_IN_TARGET])
      if any(k == 'in_hand' for k in kinds.values()):
        physics.data.qpos = qpos.reshape(np.shape(physics.data.qpos))
        with physics.suppress_physics_errors():
          physics.after_reset()
        grasp_pos = np.asarray(physics.named.data.site_xpos['grasp']).reshape(B, 3)
        grasp_mat = np.asarray(physics.named.data.site_xmat['grasp']).reshape(B, 9)
      for e in idx:
        if kinds[e] == 'in_target':
          ox, oz, oang = target[e]
        elif kinds[e] == 'in_hand':
          ox, oz = grasp_pos[e, 0], grasp_pos[e, 2]
          oang = np.pi - np.arctan2(grasp_mat[e, 6], grasp_mat[e, 0])      # (zx, xx)
        else:
          ox, oz, oang = uniform(-.5, .5), uniform(0, .7), uniform(0, 2 * np.pi)
          qvel[e, m.jnt_dofadr[jid(self._object + '_x')]] = uniform(-5, 5)
        for n, v in zip(self._object_joints, (ox, oz, oang)):
          qpos[e, qadr(n)] = v
      physics.data.qpos = qpos.reshape(np.shape(physics.data.qpos))
      physics.data.qvel = qvel.reshape(np.shape(physics.data.qvel))
      with physics.suppress_physics_errors():     # a rejected sample may overflow the contact cap
        physics.after_reset()
      todo &= np.atleast_1d(physics.data.ncon) > 0
    physics.target_pose = target[0] if B == 1 else target
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['arm_pos'] = physics.bounded_joint_pos(_ARM_JOINTS)
    obs['arm_vel'] = physics.joint_vel(_ARM_JOINTS)
    obs['touch'] = physics.touch()
    if self._fully_observable:
      obs['hand_pos'] = physics.body_2d_pose('hand')
      obs['object_pos'] = physics.body_2d_pose(self._object)
      obs['object_vel'] = physics.joint_vel(self._object_joints)
      obs['target_pos'] = physics.target_2d_pose(self._target)
    return obs

  def _is_close(self, distance):
    return rewards.tolerance(distance, (0, _CLOSE), _CLOSE * 2)

  def _site_local(self, physics, site):
    return physics.model.site_pos[physics.model.name2id(site, 'site')]

  def get_reward(self, physics):
    if not self._use_peg:
      return self._is_close(physics.target_site_distance('ball', self._target, self._site_local(physics, 'target_ball')))
    grasp = self._is_close(physics.site_distance('peg_grasp', 'grasp'))
    pinch = self._is_close(physics.site_distance('peg_pinch', 'pinch'))
    grasping = (grasp + pinch) / 2
    bring = self._is_close(physics.target_site_distance('peg', self._target, self._site_local(physics, 'target_peg')))
    bring_tip = self._is_close(physics.target_site_distance('peg_tip', self._target, self._site_local(physics, 'target_peg_tip')))
    bringing = (bring + bring_tip) / 2
    return np.maximum(bringing, grasping / 3)
