"""Batch-aware composer environment loop on device.

Mirrors `composer.Environment.step / _substep` (dm_control/composer/environment.py:412-465) and the hook dispatcher
`_EnvironmentHooks` (:74-162) for B environments at once:

    before_step(physics, action)            -- task, then entities
    for each of n_sub_steps:
      before_substep -> physics.step() -> after_substep
    after_step
    reward / discount / should_terminate_episode or time limit

Every hook receives the batch physics and works on (.., B) tensors; per-env quantities the reference keeps as Python
scalars on the task (`_failure_termination`, `_reward_step_counter`) are (B,) tensors.  Like the reference, hooks that
are no-ops are found once at construction (`_callable_is_trivial`, environment.py:44-58) and never called; when no
substep hook is left, the n_sub_steps physics steps of a control step fuse into ONE launch of the step kernel
(`Physics.step(n_sub_steps)`), otherwise each substep is its own launch with the hooks in between, in the
reference's order.  `legacy_base.Walker.after_substep` (legacy_base.py:179-186: `mj_subtreeVel`, needed so that
`subtree_linvel` is fresh) has no work left here: the kernel evaluates subtree velocities whenever a
`subtreelinvel` sensor or output asks for them.

An environment whose episode ended is re-initialised at the start of the NEXT `step` call, which returns its FIRST
time step (reward / discount of that step are zero / one and `step_type == 0`), as the reference does with
`_reset_next_step` (environment.py:414-416) -- per environment.  Nothing in the steady state synchronises with the
host.
"""
import collections

import numpy as np

HOOK_NAMES = ('initialize_episode', 'before_step', 'before_substep', 'after_substep', 'after_step')
FIRST, MID, LAST = 0, 1, 2

TimeStep = collections.namedtuple('TimeStep', ['step_type', 'reward', 'discount', 'observation'])


def _empty_function():
  pass


def _empty_function_with_docstring():
  """Some docstring."""


_EMPTY_CODE = _empty_function.__code__.co_code
_EMPTY_WITH_DOCSTRING_CODE = _empty_function_with_docstring.__code__.co_code


def _callable_is_trivial(f):
  """environment.py:44-58: a hook whose body is empty (or only a docstring) is never dispatched."""
  try:
    code = f.__code__.co_code
  except AttributeError:
    return False
  return code in (_EMPTY_CODE, _EMPTY_WITH_DOCSTRING_CODE)


class Entity:
  """The slice of `composer.Entity` the loop needs: named, with optional hooks (all trivial by default)."""

  def initialize_episode(self, physics, random_state, mask):
    pass

  def before_step(self, physics, random_state):
    pass

  def before_substep(self, physics, random_state):
    pass

  def after_substep(self, physics, random_state):
    pass

  def after_step(self, physics, random_state):
    pass


class Task:
  """`composer.Task` for a batch (composer/task.py): hooks default to no-ops; subclasses override what they use.

  Differences forced by the batch: `initialize_episode` gets the (B,) bool mask of the environments being
  re-initialised; `get_reward` / `get_discount` / `should_terminate_episode` return (B,) tensors (rewards may be
  (n_agents, B))."""

  physics_timestep = 0.005
  control_timestep = 0.025
  restarting = None      # set by Environment.step: (B,) mask of the environments re-initialised in this call

  @property
  def entities(self):
    return ()

  def set_timesteps(self, control_timestep, physics_timestep):
    n = control_timestep / physics_timestep
    if abs(n - round(n)) > 1e-6:
      raise ValueError('control timestep must be an integer multiple of the physics timestep')
    self.control_timestep, self.physics_timestep = control_timestep, physics_timestep

  @property
  def physics_steps_per_control_step(self):
    return int(round(self.control_timestep / self.physics_timestep))

  def initialize_episode(self, physics, random_state, mask):
    pass

  def before_step(self, physics, action, random_state):
    pass

  def before_substep(self, physics, action, random_state):
    pass

  def after_substep(self, physics, random_state):
    pass

  def after_step(self, physics, random_state):
    pass

  def get_reward(self, physics):
    raise NotImplementedError

  def get_discount(self, physics):
    return physics.torch.ones(physics.B, dtype=physics.dtype, device=physics.device)

  def should_terminate_episode(self, physics):
    return physics.torch.zeros(physics.B, dtype=physics.torch.bool, device=physics.device)

  def get_observation(self, physics):
    raise NotImplementedError


class _EnvironmentHooks:
  """environment.py:74-162: scans the task and its entities once, keeps the non-trivial hooks."""

  def __init__(self, task):
    self._task = task
    self._entity_hooks = {n: [] for n in HOOK_NAMES}
    self._extra = {n: [] for n in HOOK_NAMES}
    self._task_hooks = {}
    # Entities whose after_substep hook only looks at ONE geom's position can be served from the step kernel's substep
    # probe (DevicePhysics.substep_probe): they expose `substep_probe_geom` and `after_substeps(physics, trace)`, the
    # control step stays one launch, and their detections are those of the per-substep hook (position_detector.py's
    # retain_substep_detections semantics).  Used when every one of them names the same geom; else the hooks run.
    probed = [e for e in task.entities if getattr(e, 'substep_probe_geom', None) is not None and hasattr(e, 'after_substeps')]
    geoms = {e.substep_probe_geom for e in probed}
    self.probe_geom = geoms.pop() if len(geoms) == 1 else None
    self.probe_entities = probed if self.probe_geom is not None else []
    self._hook_entities = {}
    for name in HOOK_NAMES:
      h = getattr(task, name)
      self._task_hooks[name] = None if _callable_is_trivial(h) else h
      for entity in task.entities:
        eh = getattr(entity, name, None)
        if eh is not None and not _callable_is_trivial(eh):
          self._entity_hooks[name].append(eh)
          self._hook_entities[eh] = entity

  def add_extra_hook(self, hook_name, hook_callable):
    if hook_name not in HOOK_NAMES:
      raise ValueError('{!r} is not a valid hook name'.format(hook_name))
    if not callable(hook_callable):
      raise ValueError('{!r} is not a callable'.format(hook_callable))
    self._extra[hook_name].append(hook_callable)

  def has_substep_hooks(self, probed_count=False):
    """probed_count: also count the after_substep hooks the substep probe can serve."""
    def left(n):
      hooks = self._entity_hooks[n]
      if n == 'after_substep' and not probed_count:
        hooks = [h for h in hooks if self._hook_entities[h] not in self.probe_entities]
      return hooks
    return any(self._task_hooks[n] or left(n) or self._extra[n] for n in ('before_substep', 'after_substep'))

  def initialize_episode(self, physics, random_state, mask):
    if self._task_hooks['initialize_episode']:
      self._task_hooks['initialize_episode'](physics, random_state, mask)
    for h in self._entity_hooks['initialize_episode']:
      h(physics, random_state, mask)
    for h in self._extra['initialize_episode']:
      h(physics, random_state, mask)

  def before_step(self, physics, action, random_state):
    if self._task_hooks['before_step']:
      self._task_hooks['before_step'](physics, action, random_state)
    for h in self._entity_hooks['before_step']:
      h(physics, random_state)
    for h in self._extra['before_step']:
      h(physics, action, random_state)

  def before_substep(self, physics, action, random_state):
    if self._task_hooks['before_substep']:
      self._task_hooks['before_substep'](physics, action, random_state)
    for h in self._entity_hooks['before_substep']:
      h(physics, random_state)
    for h in self._extra['before_substep']:
      h(physics, action, random_state)

  def after_substep(self, physics, random_state):
    if self._task_hooks['after_substep']:
      self._task_hooks['after_substep'](physics, random_state)
    for h in self._entity_hooks['after_substep']:
      h(physics, random_state)
    for h in self._extra['after_substep']:
      h(physics, random_state)

  def after_step(self, physics, random_state):
    if self._task_hooks['after_step']:
      self._task_hooks['after_step'](physics, random_state)
    for h in self._entity_hooks['after_step']:
      h(physics, random_state)
    for h in self._extra['after_step']:
      h(physics, random_state)


class Environment:
  """B composer-style environments of one task on one GPU."""

  def __init__(self, task, physics, time_limit=float('inf'), random_state=None, n_sub_steps=None, fuse_substeps=None,
               observation_options=None, delayed_observation_padding='zero', strip_singleton_obs_buffer_dim=True):
    """observation_options: {observable name: dict(update_interval=, buffer_size=, delay=, aggregator=)} -- the attributes the
    reference sets on `task.observables[...]` / `entity.observables.<name>` (observable/base.py:55-63), in physics steps;
    `delayed_observation_padding` ('zero' / 'initial_value') and `strip_singleton_obs_buffer_dim` are
    `composer.Environment`'s arguments of the same names (composer/environment.py:188-196; the reference's default for the
    latter is False -- this class has always returned (B, n), which is True).  See composer/updater.py."""
    self.task = task
    self.physics = physics
    self._time_limit = time_limit
    self._rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    self._n_sub_steps = n_sub_steps or task.physics_steps_per_control_step
    self._hooks = _EnvironmentHooks(task)
    self._fuse = fuse_substeps
    torch = physics.torch
    self._reset_next = torch.ones(physics.B, dtype=torch.bool, device=physics.device)
    self._host_all_reset = True     # known without a device sync: every env is waiting for its reset
    self.launches = 0               # physics STEP launches issued (tests / profiling)
    self.forward_launches = 0       # mj_forward launches ahead of the observations (observation_forward tasks)
    self._updater = None
    if observation_options or delayed_observation_padding != 'zero' or not strip_singleton_obs_buffer_dim:
      from dm_control_amd.composer import updater
      self._updater = updater.Updater(torch, physics.B, self._n_sub_steps, observation_options,
                                      pad=delayed_observation_padding, strip_singleton_buffer_dim=strip_singleton_obs_buffer_dim)

  @property
  def n_sub_steps(self):
    return self._n_sub_steps

  @property
  def fused(self):
    return (not self._hooks.has_substep_hooks()) if self._fuse is None else bool(self._fuse)

  @property
  def probed(self):
    """The fused launch serves position-only after_substep hooks from the substep probe (their per-substep semantics
    are kept); `fuse_substeps=True` forces fusion WITHOUT it (the hooks then see the control step as one substep)."""
    return self._fuse is None and self.fused and bool(self._hooks.probe_entities)

  def add_extra_hook(self, hook_name, hook_callable):
    self._hooks.add_extra_hook(hook_name, hook_callable)

  def control_timestep(self):
    return self.task.physics_timestep * self._n_sub_steps

  # -- episode initialisation (environment.py:366-395) ------------------------------------------------
  def _initialize(self, mask):
    """`with physics.reset_context(): hooks.initialize_episode(...)` for the masked environments: mj_resetData,
    then the task's initialisation; the closing mj_forward with actuation disabled (engine.py:326-333) is the
    caller's launch."""
    p = self.physics
    p.reset(mask)
    self._hooks.initialize_episode(p, self._rs, mask)

  def reset(self):
    torch = self.physics.torch
    mask = torch.ones(self.physics.B, dtype=torch.bool, device=self.physics.device)
    self._initialize(mask)
    self.physics.field('env_mode').zero_()
    self.physics.forward(disable_actuation=True)
    self.launches += 1
    self._reset_next.zero_()
    self._host_all_reset = False
    B = self.physics.B
    obs = self.task.get_observation(self.physics)
    if self._updater is not None:
      obs = self._updater.corrupt(obs)
      self._updater.start(obs, mask)      # Updater.reset: the first sample, at time 0
      obs = self._updater.read(obs)
    return TimeStep(step_type=torch.full((B,), FIRST, dtype=torch.int32, device=self.physics.device),
                    reward=torch.zeros(B, dtype=self.physics.dtype, device=self.physics.device),
                    discount=torch.ones(B, dtype=self.physics.dtype, device=self.physics.device),
                    observation=obs)

  # -- one control step (environment.py:412-465) --------------------------------------------------------
  def step(self, action):
    p, torch = self.physics, self.physics.torch
    if self._host_all_reset:
      return self.reset()
    # a task may offer the whole control step as a few launches of its own kernels (tasks/soccer.py); None = not here
    device_step = getattr(self.task, 'device_step', None) if self._updater is None else None
    if device_step is not None:
      ts = device_step(self, action)
      if ts is not None:
        return ts
    first = self._reset_next.clone()
    # environments whose episode ended last step start a new one now; the others take a regular step.  The
    # re-initialisation is data-dependent but needs no host decision: the state edits are applied under the mask,
    # and the SAME launch that steps the others runs mj_forward with actuation disabled for these (env_mode 1).
    self._initialize(first)
    p.field('env_mode').copy_(first[None, :].to(torch.int32))
    # The reference returns reset() for such an environment without calling before_step / get_reward / after_step
    # (environment.py:412-420): its ctrl is mj_resetData's zero, not the discarded action, and the task's carried
    # counters do not see this call (`task.restarting`: the mask of environments being re-initialised).
    task = self.task
    task.restarting = first
    self._hooks.before_step(p, action, self._rs)
    ctrl = p.field('ctrl')
    if ctrl.numel():
      ctrl.masked_fill_(first[None, :], 0)
    # observation_forward tasks: the launch that ends the control step also runs the rest of mj_forward at the new state
    # (dmc_batch_step legacy_step 2) -- see below; environments re-initialised in this call are not touched by it
    obs_forward = bool(getattr(task, 'observation_forward', False))
    fold = obs_forward and getattr(p, 'supports_forward_after', True)
    upd = self._updater
    running = ~first
    substeps = upd is not None and upd.needs_substeps      # an observable is sampled between the physics steps
    if substeps:
      # updater.py:292-301 after every substep (composer/environment.py:455-458); the last one's pass is the closing one below
      for k in range(self._n_sub_steps):
        self._hooks.before_substep(p, action, self._rs)
        p.step(1, forward_after=fold and k == self._n_sub_steps - 1)
        self.launches += 1
        self._hooks.after_substep(p, self._rs)
        upd.advance(1, running)
        if k < self._n_sub_steps - 1:
          upd.sample(upd.corrupt(task.get_observation(p)), running)
    elif self.probed:
      # one launch; the kernel leaves the probed geom's position after every substep, the entities read that trace
      trace = p.substep_probe(self._hooks.probe_geom, self._n_sub_steps)
      p.step(self._n_sub_steps, forward_after=fold)
      self.launches += 1
      for e in self._hooks.probe_entities:
        e.after_substeps(p, trace[:self._n_sub_steps])
    elif self.fused:
      # no substep hook (or fusion forced): the substep hooks, if any, see the control step as ONE substep
      self._hooks.before_substep(p, action, self._rs)
      p.step(self._n_sub_steps, forward_after=fold)
      self.launches += 1
      self._hooks.after_substep(p, self._rs)
    else:
      for k in range(self._n_sub_steps):
        self._hooks.before_substep(p, action, self._rs)
        p.step(1, forward_after=fold and k == self._n_sub_steps - 1)
        self.launches += 1
        self._hooks.after_substep(p, self._rs)
    if upd is not None and not substeps:
      upd.advance(self._n_sub_steps, running)
    self._hooks.after_step(p, self._rs)
    if obs_forward and not fold:
      # In the reference the action reaches mjData through an mjcf binding (walker.apply_action), which marks the physics
      # dirty; physics.step() does not clear that, so the first observable read through a binding after the substeps
      # runs mj_forward (mjcf/physics.py:341-342): the acceleration-stage sensors an agent sees (touch, torque,
      # accelerometer, force) are those of the NEW state under the action just applied, not the last substep's.
      # Found by running composer/environment.py unmodified on this backend (tests/test_reference_composer.py).
      # The environments re-initialised in this call keep their mj_forward with actuation disabled (env_mode 2: untouched).
      p.field('env_mode').copy_((first.to(torch.int32) * 2)[None, :])
      p.forward()
      self.forward_launches += 1
    reward = task.get_reward(p)
    discount = task.get_discount(p)
    # ANY new mjWARN_* since the last look = PhysicsError in the reference (engine.py:345-368 compares the warning
    # counters before / after; BADQPOS / BADQVEL / BADQACC but also CONTACTFULL / CNSTRFULL / INERTIA): reward 0,
    # discount 0, episode over (environment.py:449-452)
    diverged = self._divergence(p)
    terminating = task.should_terminate_episode(p) | (p.field('time')[0] >= self._time_limit - 1e-9) | diverged
    zero = torch.zeros_like(discount)
    reward = torch.where(diverged if reward.dim() == 1 else diverged[None, :], torch.zeros_like(reward), reward)
    discount = torch.where(diverged, zero, discount)
    obs = task.get_observation(p)
    if upd is not None:
      obs = upd.corrupt(obs)
      upd.sample(obs, running)      # the pass after the control step's last physics step
      upd.start(obs, first)         # environments re-initialised in this call: Updater.reset, the first sample of the new episode
      obs = upd.read(obs)
    step_type = torch.where(first, torch.full_like(first, FIRST, dtype=torch.int32),
                            torch.where(terminating, torch.full_like(first, LAST, dtype=torch.int32),
                                        torch.full_like(first, MID, dtype=torch.int32)))
    # an environment that was (re)started this call reports FIRST with no reward, and cannot end on it
    reward = torch.where(first if reward.dim() == 1 else first[None, :], torch.zeros_like(reward), reward)
    discount = torch.where(first, torch.ones_like(discount), discount)
    self._reset_next.copy_(terminating & ~first)      # in place: carried state must keep its address (graph replay)
    return TimeStep(step_type=step_type, reward=reward, discount=discount, observation=obs)

  # -- HIP graph: one control step as ONE replayable graph ----------------------------------------------------
  def capture(self, example_action):
    """Captures `step` -- hooks, the physics launch(es), reward, termination, observation gather: a few hundred
    small device operations with no host decision in between -- into a HIP graph (torch.cuda.CUDAGraph; the
    library's kernels are launched on the capturing stream like any other).  `step_graph(action)` then replays it:
    the per-operation launch overhead, which dominates small batches (soccer at 256 environments: 10 ms of host
    time around 3.4 ms of physics), is paid once.  Every tensor the step carries over (reset flags, task state,
    detector state) is updated in place, so a replay continues where the previous one ended."""
    torch = self.physics.torch
    # (a graph keeps the kernel it was captured with: a specialised kernel still compiling in the background is waited for)
    getattr(getattr(self.physics, 'batch', None), 'wait_specialised', lambda: None)()
    if self._host_all_reset:
      self.reset()
    self._g_action = example_action.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):              # warm-up off the default stream, as graph capture requires
      for _ in range(2):
        self.step(self._g_action)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    for gen in getattr(self.task, 'generators', lambda: ())():
      graph.register_generator_state(gen)
    with torch.cuda.graph(graph):
      self._g_out = self.step(self._g_action)
    self._graph = graph
    return self._g_out

  def step_graph(self, action):
    """Replays the captured control step on `action`; returns the SAME TimeStep tensors, refreshed."""
    self._g_action.copy_(action)
    self._graph.replay()
    return self._g_out

  def _divergence(self, physics):
    w = physics.field('warning')
    bad = w[:8].sum(dim=0) > 0                # the eight mjtWarning counters (include/dmc_model_layout.h order); row 8 is this backend's own
    w[:8].zero_()
    return bad

  def close(self):
    self.physics.close()
