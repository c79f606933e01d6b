"""RL environment loop over a physics backend.

Host-side mirror of the reference's plug-in seam, `dm_control/rl/control.py`:
`Environment` (:28-165: reset :82, step :99), `compute_n_steps` (:168), the
`Physics` ABC (:206-267), `PhysicsError` (:270), the `Task` ABC (:274-371) and
`flatten_observation` (:374).  Same names, argument meaning and error
behaviour, so suite-style tasks run unchanged; additionally every method is
batch-aware: with a batched `Physics` (batch_size B > 1) observations / rewards
carry a leading batch dimension and the episode bookkeeping is shared (all
environments of a batch run in lock-step, auto-reset per environment is the
task's choice).
"""
import abc
import collections
import contextlib

import numpy as np

from dm_control_amd.envs import dm_env_api as dm_env
from dm_control_amd.envs.dm_env_api import specs

FLAT_OBSERVATION_KEY = 'observations'


class PhysicsError(RuntimeError):
  """Raised if the state of the physics simulation becomes divergent."""


class Physics(metaclass=abc.ABCMeta):
  """Simulates a physical environment (abstract)."""

  legacy_step = True

  @abc.abstractmethod
  def step(self, n_sub_steps=1):
    pass

  @abc.abstractmethod
  def time(self):
    pass

  @abc.abstractmethod
  def timestep(self):
    pass

  def set_control(self, control):
    raise NotImplementedError('set_control is not supported.')

  @contextlib.contextmanager
  def reset_context(self):
    """`with physics.reset_context(): <set state>` -- reset() on entry (a
    PhysicsError there is swallowed), after_reset() on exit."""
    try:
      self.reset()
    except PhysicsError:
      pass
    yield self
    self.after_reset()

  @abc.abstractmethod
  def reset(self):
    pass

  @abc.abstractmethod
  def after_reset(self):
    pass

  def check_divergence(self):
    """Raises PhysicsError if the state is divergent; default: no-op."""


class Task(metaclass=abc.ABCMeta):
  """Defines a task in a `control.Environment` (abstract)."""

  @abc.abstractmethod
  def initialize_episode(self, physics):
    pass

  @abc.abstractmethod
  def before_step(self, action, physics):
    pass

  def after_step(self, physics):
    pass

  @abc.abstractmethod
  def action_spec(self, physics):
    pass

  def step_spec(self, physics):
    raise NotImplementedError()

  @abc.abstractmethod
  def get_observation(self, physics):
    pass

  @abc.abstractmethod
  def get_reward(self, physics):
    pass

  def get_termination(self, physics):
    """None while the episode continues, else the final discount."""

  @abc.abstractmethod
  def observation_spec(self, physics):
    pass


def compute_n_steps(control_timestep, physics_timestep, tolerance=1e-8):
  """Number of physics steps per control step; ValueError if not an integer
  multiple or smaller than the physics step."""
  if control_timestep < physics_timestep:
    raise ValueError('Control timestep ({}) cannot be smaller than physics timestep ({}).'.format(
        control_timestep, physics_timestep))
  ratio = control_timestep / physics_timestep
  if abs(ratio - round(ratio)) > tolerance:
    raise ValueError('Control timestep ({}) must be an integer multiple of physics timestep ({})'.format(
        control_timestep, physics_timestep))
  return int(round(ratio))


def flatten_observation(observation, output_key=FLAT_OBSERVATION_KEY):
  """Flattens a dict of arrays into {output_key: 1-D array}; key order is
  preserved for ordered dicts and sorted otherwise."""
  if not isinstance(observation, collections.abc.MutableMapping):
    raise ValueError('Can only flatten dict-like observations.')
  keys = observation.keys() if isinstance(observation, collections.OrderedDict) else sorted(observation.keys())
  flat = np.concatenate([np.ravel(observation[k]) for k in keys])
  return type(observation)([(output_key, flat)])


def _spec_from_observation(observation):
  out = collections.OrderedDict()
  for key, value in observation.items():
    value = np.asarray(value)
    out[key] = specs.Array(value.shape, value.dtype, name=key)
  return out


class Environment(dm_env.Environment):
  """The agent-facing loop of the reference (rl/control.py:40-176) over any `control.Physics`: a control step is
  task.before_step -> physics.step(n_sub_steps) -> task.after_step, then reward, observation and the episode-end
  test (time budget first, then the task's termination).  Works unchanged for a batch physics: the task then
  returns (B,) rewards and (B, ...) observations."""

  def __init__(self, physics, task, time_limit=float('inf'), control_timestep=None,
               n_sub_steps=None, flat_observation=False, legacy_step=True):
    if n_sub_steps is not None and control_timestep is not None:
      raise ValueError('Both n_sub_steps and control_timestep were supplied.')
    physics.legacy_step = legacy_step
    if n_sub_steps is None:
      n_sub_steps = 1 if control_timestep is None else compute_n_steps(control_timestep, physics.timestep())
    self._physics, self._task = physics, task
    self._n_sub_steps = n_sub_steps
    self._flat_observation = flat_observation
    # control steps an episode may take (fractional: the comparison below is >=)
    self._step_limit = time_limit if time_limit == float('inf') else time_limit / (physics.timestep() * n_sub_steps)
    self._step_count = 0
    self._reset_next_step = True

  def _observe(self):
    obs = self._task.get_observation(self._physics)
    return flatten_observation(obs) if self._flat_observation else obs

  def reset(self):
    self._step_count, self._reset_next_step = 0, False
    with self._physics.reset_context():
      self._task.initialize_episode(self._physics)
    return dm_env.restart(self._observe())

  def step(self, action):
    if self._reset_next_step:      # the call after an episode's last step starts the next episode
      return self.reset()
    task, physics = self._task, self._physics
    task.before_step(action, physics)
    physics.step(self._n_sub_steps)
    task.after_step(physics)
    self._step_count += 1
    reward, observation = task.get_reward(physics), self._observe()
    if isinstance(reward, np.ndarray) and reward.ndim == 0:
      reward = reward[()]      # a single environment's reward is a (numpy) float, as the reference's tasks return it
    final_discount = 1.0 if self._step_count >= self._step_limit else task.get_termination(physics)
    if final_discount is None:
      return dm_env.transition(reward, observation)
    self._reset_next_step = True
    return dm_env.TimeStep(dm_env.StepType.LAST, reward, final_discount, observation)

  def action_spec(self):
    return self._task.action_spec(self._physics)

  def step_spec(self):
    return self._task.step_spec(self._physics)

  def observation_spec(self):
    try:
      return self._task.observation_spec(self._physics)
    except NotImplementedError:
      return _spec_from_observation(self._observe())

  @property
  def physics(self):
    return self._physics

  @property
  def task(self):
    return self._task

  def control_timestep(self):
    return self._physics.timestep() * self._n_sub_steps
