"""`suite.load(domain, task)` over the HIP physics backend.

Mirrors dm_control/suite/__init__.py:93-150 (`load`, `build_environment`,
ALL_TASKS / BENCHMARKING) for the domains whose models the BASELINE configs
name (cartpole, cheetah, humanoid) plus the domains that share their feature set
(acrobot, ball_in_cup, finger, fish, hopper, humanoid_CMU, lqr, manipulator, pendulum, point_mass, quadruped, reacher, stacker, swimmer, walker).  Extra keyword: `physics_kwargs`
(batch_size, precision, device_id, ...) to run a whole batch behind the same
`Environment` API.
"""
import collections

from dm_control_amd.suite import acrobot
from dm_control_amd.suite import ball_in_cup
from dm_control_amd.suite import cartpole
from dm_control_amd.suite import cheetah
from dm_control_amd.suite import finger
from dm_control_amd.suite import fish
from dm_control_amd.suite import hopper
from dm_control_amd.suite import humanoid
from dm_control_amd.suite import humanoid_CMU
from dm_control_amd.suite import lqr
from dm_control_amd.suite import manipulator
from dm_control_amd.suite import pendulum
from dm_control_amd.suite import point_mass
from dm_control_amd.suite import quadruped
from dm_control_amd.suite import reacher
from dm_control_amd.suite import stacker
from dm_control_amd.suite import swimmer
from dm_control_amd.suite import walker

_DOMAINS = collections.OrderedDict(acrobot=acrobot, ball_in_cup=ball_in_cup, cartpole=cartpole, cheetah=cheetah, finger=finger, fish=fish,
                                   hopper=hopper,
                                   humanoid=humanoid, humanoid_CMU=humanoid_CMU, lqr=lqr, manipulator=manipulator, pendulum=pendulum, point_mass=point_mass, quadruped=quadruped, reacher=reacher,
                                   stacker=stacker, swimmer=swimmer, walker=walker)

ALL_TASKS = tuple((d, t) for d, mod in _DOMAINS.items() for t in mod.TASKS)
BENCHMARKING = tuple((d, t) for d, mod in _DOMAINS.items() for t, (_, tag) in mod.TASKS.items()
                     if tag == 'benchmarking')
# difficulty tags of the reference's `@SUITE.add(...)` decorators (suite/__init__.py:79-86)
EASY = (('ball_in_cup', 'catch'), ('point_mass', 'easy'), ('reacher', 'easy'))
HARD = tuple((d, t) for d, t in ALL_TASKS if d in ('manipulator', 'stacker'))
EXTRA = tuple(sorted(set(ALL_TASKS) - set(BENCHMARKING)))
NO_REWARD_VIZ = ()                       # the dog tasks in the reference; none of them is provided here
REWARD_VIZ = tuple(sorted(ALL_TASKS))
TASKS_BY_DOMAIN = collections.OrderedDict((d, tuple(mod.TASKS)) for d, mod in _DOMAINS.items())


def build_environment(domain_name, task_name, task_kwargs=None, environment_kwargs=None, physics_kwargs=None):
  if domain_name not in _DOMAINS:
    raise ValueError('Domain {!r} does not exist.'.format(domain_name))
  domain = _DOMAINS[domain_name]
  if task_name not in domain.TASKS:
    raise ValueError('Level {!r} does not exist in domain {!r}.'.format(task_name, domain_name))
  kwargs = dict(task_kwargs or {})
  if environment_kwargs is not None:
    kwargs['environment_kwargs'] = dict(environment_kwargs)
  if physics_kwargs is not None:
    kwargs['physics_kwargs'] = dict(physics_kwargs)
  env = domain.TASKS[task_name][0](**kwargs)
  env.task.visualize_reward = False
  return env


def load(domain_name, task_name, task_kwargs=None, environment_kwargs=None, visualize_reward=False,
         physics_kwargs=None):
  del visualize_reward  # rendering-only
  return build_environment(domain_name, task_name, task_kwargs, environment_kwargs, physics_kwargs)
