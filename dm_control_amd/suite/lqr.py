"""LQR domain (reference: dm_control/suite/lqr.py): lqr_2_1, lqr_6_2.

A chain of n masses on springs (random stiffness in [15, 25]) of which the first m are
actuated; cost = sum(states^2) + c sum(controls^2).  The model is generated from the
template as the reference does (lqr.py:131-187); the connecting spatial tendons are
force-free rendering aids.  With a batch, the episode terminates when EVERY environment
has converged (the host Environment takes one termination decision per step)."""
import collections
import xml.etree.ElementTree as etree

import numpy as np

from dm_control_amd import physics as physics_lib
from dm_control_amd.envs import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common

_DEFAULT_TIME_LIMIT = float('inf')
_CONTROL_COST_COEF = 0.1
TASKS = {}


def get_model_and_assets(n_bodies, n_actuators, random):
  return _make_model(n_bodies, n_actuators, random), None


def _make_model(n_bodies, n_actuators, random, stiffness_range=(15, 25), damping_range=(0, 0)):
  if n_bodies < 1 or n_actuators < 1:
    raise ValueError('At least 1 body and 1 actuator required.')
  if n_actuators > n_bodies:
    raise ValueError('At most 1 actuator per body.')
  mjcf = etree.fromstring(common.read_model('lqr.xml'))
  parent = mjcf.find('./worldbody')
  actuator = etree.SubElement(mjcf, 'actuator')
  tendon = etree.SubElement(mjcf, 'tendon')
  for body in range(n_bodies):
    child = etree.Element('body', name='body_{}'.format(body), pos='.25 0 0')
    joint = etree.SubElement(child, 'joint', name='joint_{}'.format(body))
    etree.SubElement(child, 'geom', name='geom_{}'.format(body))
    joint.set('stiffness', str(random.uniform(stiffness_range[0], stiffness_range[1])))
    joint.set('damping', str(random.uniform(damping_range[0], damping_range[1])))
    etree.SubElement(child, 'site', name='site_{}'.format(body))
    if body == 0:
      child.set('pos', '.25 0 .1')
    if body < n_actuators:
      etree.SubElement(actuator, 'motor', name='motor_{}'.format(body), joint='joint_{}'.format(body))
    if body < n_bodies - 1:
      spatial = etree.SubElement(tendon, 'spatial', name='tendon_{}'.format(body))
      etree.SubElement(spatial, 'site', site='site_{}'.format(body))
      etree.SubElement(spatial, 'site', site='site_{}'.format(body + 1))
    parent.append(child)
    parent = child
  return etree.tostring(mjcf, encoding='unicode')


def _make_lqr(n_bodies, n_actuators):
  def factory(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None, physics_kwargs=None):
    if not isinstance(random, np.random.RandomState):
      random = np.random.RandomState(random)
    physics = Physics.from_xml_string(*get_model_and_assets(n_bodies, n_actuators, random=random),
                                      **(physics_kwargs or {}))
    return control.Environment(physics, LQRLevel(_CONTROL_COST_COEF, random=random), time_limit=time_limit,
                               **(environment_kwargs or {}))
  return factory


lqr_2_1, lqr_6_2 = _make_lqr(2, 1), _make_lqr(6, 2)
TASKS.update(lqr_2_1=(lqr_2_1, None), lqr_6_2=(lqr_6_2, None))


class Physics(physics_lib.Physics):

  def state_norm(self):
    return common.vnorm(self.get_state())


class LQRLevel(base.Task):
  _TERMINAL_TOL = 1e-6

  def __init__(self, control_cost_coef, random=None):
    if control_cost_coef <= 0:
      raise ValueError('control_cost_coef must be positive.')
    self._control_cost_coef = control_cost_coef
    super().__init__(random=random)

  @property
  def control_cost_coef(self):
    return self._control_cost_coef

  def initialize_episode(self, physics):
    """Random state on the sphere of radius sqrt(2)."""
    B, ndof = physics.batch_size, physics.model.nq
    if B == 1:
      unit = self.random.randn(ndof)
      q = np.sqrt(2) * unit / np.linalg.norm(unit)
    else:
      unit = self.random.randn(B, ndof)
      q = np.sqrt(2) * unit / np.linalg.norm(unit, axis=-1, keepdims=True)
    physics.data.qpos = q
    super().initialize_episode(physics)

  def get_observation(self, physics):
    obs = collections.OrderedDict()
    obs['position'] = physics.position()
    obs['velocity'] = physics.velocity()
    return obs

  def get_reward(self, physics):
    position = physics.position()
    state_cost = 0.5 * common.vdot(position, position)
    u = physics.control()
    control_l2_norm = 0.5 * common.vdot(u, u)
    return 1 - (state_cost + control_l2_norm * self._control_cost_coef)

  def get_evaluation(self, physics):
    return common.asarray(physics.state_norm() <= 0.01, dtype=np.float64)

  def get_termination(self, physics):
    if np.all(physics.state_norm() < self._TERMINAL_TOL):
      return 0.0
    return None

  # per-environment form for the device-resident environments (suite/fused_env.py): where the reference's test
  # (suite/lqr.py:264: `if physics.state_norm() < _TERMINAL_TOL: return 0.0`) holds, that environment's episode ends with
  # discount `termination_discount` -- each environment at its own step, not when the whole batch has converged
  termination_discount = 0.0

  def termination_mask(self, physics):
    return physics.state_norm() < self._TERMINAL_TOL
