"""Device-resident batched environments.

`control.Environment` over the host mirror costs a PCIe round trip per field per
step.  The environments here keep everything on the GPU: the SoA arrays of the HIP
batch are rebound to torch tensors (zero copy, `dmc_batch_bind`), actions are
written by the caller's policy directly into the `ctrl` tensor, observations /
rewards are torch expressions over the bound tensors, and environments whose
episode ended are re-initialised on device with a fresh draw per episode
(`dmc_batch_randomize_joints`: a counter-based Philox stream per environment and
draw, suite/utils/randomizers.py:35-88 -- cheetah: limited joints ~ U(range), then
200 settle steps as suite/cheetah.py:63-76 does; humanoid: redrawn until nothing is
in contact, suite/humanoid.py:160-165, the rejection loop running for the whole
batch at once on the environments still pending).
torch is plumbing here (memory + elementwise ops); the physics is the fused HIP
kernel.

    env = torch_env.make('cheetah', 'run', batch_size=4096)
    obs = env.reset()
    obs, reward, done = env.step(policy(obs))      # all (B, ...) tensors on the GPU
"""
import math

import numpy as np

from dm_control_amd import mjcf_compiler
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common


def tolerance(torch, x, bounds=(0.0, 0.0), margin=0.0, sigmoid='gaussian', value_at_margin=0.1):
  """rewards.tolerance (dm_control/utils/rewards.py:93-139) on torch tensors; the sigmoids the
  device tasks use: gaussian, linear, quadratic."""
  lower, upper = bounds
  in_bounds = (x >= lower) & (x <= upper)
  if margin == 0:
    return in_bounds.to(x.dtype)
  d = torch.where(x < lower, lower - x, x - upper) / margin
  if sigmoid == 'gaussian':
    s = torch.exp(-0.5 * (d * math.sqrt(-2 * math.log(value_at_margin)))**2)
  elif sigmoid == 'linear':
    sx = d * (1 - value_at_margin)
    s = torch.where(sx.abs() < 1, 1 - sx, torch.zeros_like(sx))
  elif sigmoid == 'quadratic':
    sx = d * math.sqrt(1 - value_at_margin)
    s = torch.where(sx.abs() < 1, 1 - sx**2, torch.zeros_like(sx))
  else:
    raise ValueError('sigmoid %r is not available on device' % sigmoid)
  return torch.where(in_bounds, torch.ones_like(s), s)


class TorchBatchedEnv:
  """B environments of one suite task; every tensor is (rows, B) on `device`.

  Subclasses define `_MODEL`, `_OUTPUTS` (extra derived arrays to bind), `_initialize_episode()`,
  `observation()` and `reward()`.  The default (this class) is cheetah `run`."""

  _MODEL = 'cheetah.xml'
  _OUTPUTS = ()            # e.g. ('xpos', 'xmat'): derived arrays the task reads
  _CONTROL_TIMESTEP = None  # None: one physics step per env step unless n_sub_steps is given
  _RUN_SPEED = 10.0

  def __init__(self, batch_size, device_id=0, precision=32, time_limit=10.0, seed=0, n_sub_steps=None, capture=False,
               copy_outputs=True):
    import torch
    self._capture, self._graph = bool(capture), None
    # capture=True: a replayed HIP graph writes its results into the SAME tensors every step.  copy_outputs (default)
    # hands the caller its own copies, so that obs_t kept across step t + 1 stays obs_t; copy_outputs=False returns the
    # graph's own output tensors (no copy: valid until the next step() call).
    self._copy_outputs = bool(copy_outputs)
    self.torch = torch
    self.device = torch.device('cuda', device_id)
    self.model = mjcf_compiler.compile_xml(self._model_xml())
    m = self.model
    self.B = int(batch_size)
    if n_sub_steps is None:
      n_sub_steps = 1 if self._CONTROL_TIMESTEP is None else int(round(self._CONTROL_TIMESTEP / m.opt.timestep))
    self.n_sub_steps = n_sub_steps
    self.dtype = torch.float32 if precision == 32 else torch.float64
    caps = dict(common.DEFAULT_CAPS.get(self._MODEL[:-4], {}))
    if caps.pop('precision', precision) != precision:
      raise ValueError('%s only runs in fp%d (its fp64 scratch does not fit in LDS)' % (self._MODEL, common.DEFAULT_CAPS[self._MODEL[:-4]]['precision']))
    self.physics = BatchedPhysics(m, self.B, device_id=device_id, precision=precision, **caps)
    mask = OUT['sensor']
    for name in self._OUTPUTS:
      mask |= OUT[{'subtree_com': 'subtree_com'}.get(name, name)]
    self.physics.set_output_mask(mask)
    mk = lambda rows: torch.zeros((rows, self.B), dtype=self.dtype, device=self.device)
    self.qpos, self.qvel, self.ctrl = mk(m.nq), mk(m.nv), mk(m.nu)
    self.warm, self.sensordata = mk(m.nv), mk(m.nsensordata)
    self.time = torch.zeros((1, self.B), dtype=torch.float64, device=self.device)
    self.ncon = torch.zeros((1, self.B), dtype=torch.int32, device=self.device)
    self.env_mode = torch.zeros((1, self.B), dtype=torch.int32, device=self.device)
    bound = [('qpos', self.qpos), ('qvel', self.qvel), ('ctrl', self.ctrl), ('qacc_warmstart', self.warm),
             ('sensordata', self.sensordata), ('time', self.time), ('ncon', self.ncon), ('env_mode', self.env_mode)]
    if m.na:
      self.act = mk(m.na)      # activation states (filtered servos): zero at every reset
      bound.append(('act', self.act))
    for name in self._OUTPUTS:
      rows = self.physics._rows(name)[0]
      t = mk(rows)
      setattr(self, name, t)
      bound.append((name, t))
    for name, t in bound:
      self.physics.bind(name, t.data_ptr())
    self.step_limit = int(round(time_limit / (m.opt.timestep * n_sub_steps)))
    self.steps = torch.zeros(self.B, dtype=torch.int64, device=self.device)
    self._host_steps = 0     # upper bound of `steps` known without a device sync
    self._seed = int(seed)
    self._draw = torch.zeros(self.B, dtype=torch.int32, device=self.device)      # per-environment draw counters
    self._mask_i = torch.zeros(self.B, dtype=torch.int32, device=self.device)
    self._q0 = torch.from_numpy(np.ascontiguousarray(m.qpos0, dtype=np.float64)).to(self.device, self.dtype)[:, None]
    self.reset_rounds = 0      # launches the last reset needed (rejection / raising loops)
    self._none_done = torch.zeros(self.B, dtype=torch.bool, device=self.device)
    self._setup()
    self.reset()

  def _model_xml(self):
    return common.read_model(self._MODEL)

  # -- helpers -----------------------------------------------------------------------
  def _stream(self):
    return self.torch.cuda.current_stream().cuda_stream

  def _body(self, name):
    return self.model.name2id(name, 'body')

  def _setup(self):
    """Index tensors a task needs, before the first reset."""

  def _randomize(self, mask, flags):
    """randomize_limited_and_rotational_joints for the masked environments, drawn on the device."""
    self._mask_i.copy_(mask)
    self.physics.randomize_joints(self._seed, self._draw.data_ptr(), self._mask_i.data_ptr(), flags, stream=self._stream())

  def _launch_only(self, mask, fn):
    """Runs a launch for the masked environments and leaves the others untouched (env_mode 2)."""
    torch = self.torch
    self.env_mode.copy_(torch.where(mask[None, :], torch.zeros_like(self.env_mode), torch.full_like(self.env_mode, 2)))
    fn()
    self.env_mode.zero_()

  def _zero_rows(self, t, mask):
    t.copy_(self.torch.where(mask[None, :], self.torch.zeros_like(t), t))

  # -- start states ------------------------------------------------------------------
  def _initialize_episode(self, mask):
    """Cheetah.initialize_episode for the masked environments (suite/cheetah.py:63-76): limited joints ~ U(range),
    then 200 settle steps with zero control; the other environments sit the launch out."""
    torch = self.torch
    self.qpos.copy_(torch.where(mask[None, :], self._q0, self.qpos))
    self._randomize(mask, BatchedPhysics.RAND_LIMITED)
    for t in (self.qvel, self.ctrl, self.warm):
      self._zero_rows(t, mask)
    self._launch_only(mask, lambda: self.physics.step(200, stream=self._stream()))
    self.reset_rounds = 1

  def reset(self, mask=None):
    """Re-initialises the selected environments (all if mask is None) with a fresh draw each and refreshes the
    derived arrays (mj_forward with actuation disabled, as Physics.reset does)."""
    torch = self.torch
    if mask is None:
      mask = torch.ones(self.B, dtype=torch.bool, device=self.device)
      self._host_steps = 0
    if self.model.na:
      self._zero_rows(self.act, mask)
    self._initialize_episode(mask)
    self._zero_rows(self.time, mask)
    self.steps.copy_(torch.where(mask, torch.zeros_like(self.steps), self.steps))
    self.physics.invalidate(stream=self._stream())      # qpos / qvel were edited through the bound tensors (stream-ordered with the launches)
    if self._OUTPUTS:
      # refresh the derived arrays of the environments that were reset, and of those only: the others are left
      # untouched by the launch (env_mode 2), so their acceleration-stage sensors, warm starts and observations
      # do not depend on who else was reset
      self._launch_only(mask, lambda: self.physics.forward(disable_actuation=True, stream=self._stream()))
    return self.observation()

  # -- task ----------------------------------------------------------------------------
  def observation(self):
    """(B, 17): qpos[1:] and qvel (Cheetah.get_observation)."""
    return self.torch.cat([self.qpos[1:], self.qvel], dim=0).T

  def reward(self):
    """rewards.tolerance(speed, bounds=(10, inf), margin=10, value_at_margin=0, 'linear')."""
    speed = self.sensordata[0]
    return self.torch.clamp(speed * (1.0 / self._RUN_SPEED), 0.0, 1.0)

  def _steady_step(self, action):
    """A control step in which no environment can finish: ctrl write, the physics launch, counters, reward, observation --
    device operations only, no host decision."""
    self.ctrl.copy_(action.T.to(self.dtype))
    self.physics.step(self.n_sub_steps, stream=self._stream())
    self.steps += 1
    return self.observation(), self.reward().clone(), self.steps >= self.step_limit

  def _graph_step(self, action):
    """The steady step as a HIP graph (torch.cuda.CUDAGraph; the physics kernel is launched on the capturing stream): the
    ~10 small launches of the task layer cost one replay.  Returns the graph's own output tensors, rewritten by the next
    replay."""
    torch = self.torch
    if self._graph is None:
      self.physics.wait_specialised()      # (a graph keeps the kernel it was captured with)
      self._g_action = action.clone()
      state = [self.qpos, self.qvel, self.warm, self.time, self.ctrl, self.steps] + ([self.act] if self.model.na else [])
      saved = [t.clone() for t in state]
      side = torch.cuda.Stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):              # warm-up off the default stream, as graph capture requires
        self._steady_step(self._g_action)
      torch.cuda.current_stream().wait_stream(side)
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph):
        self._g_out = self._steady_step(self._g_action)
      for t, v in zip(state, saved):             # the warm-up step is taken back: the replay below is this call's step
        t.copy_(v)
      self._graph = graph
    self._g_action.copy_(action)
    self._graph.replay()
    return self._g_out

  def step(self, action):
    """action: (B, nu) tensor on device.  Returns (obs, reward, done) tensors; finished
    environments are auto-reset (their returned obs is the fresh start state).  With `capture=True, copy_outputs=False`
    the returned tensors are the HIP graph's own outputs, which the NEXT step() overwrites: clone what must outlive it."""
    if self._capture and self._host_steps + 1 < self.step_limit:      # nobody can reach the limit in this step
      out = self._graph_step(action)
      self._host_steps += 1
      return tuple(t.clone() for t in out) if self._copy_outputs else out
    self.ctrl.copy_(action.T.to(self.dtype))
    self.physics.step(self.n_sub_steps, stream=self._stream())
    self.steps += 1
    self._host_steps += 1
    reward = self.reward()
    # (no environment has taken more steps than the host-side count: below the limit `done` is all-false without asking)
    if self._host_steps >= self.step_limit:
      reward = reward.clone()      # (a reset below may rewrite what the reward is a view of)
      done = self.steps >= self.step_limit
    else:
      done = self._none_done
    obs = self.observation()
    # The time limit is the only termination, and no environment has taken more steps than the
    # host-side count since the last reset of everything: before that count reaches the limit
    # nothing can be done, so the device is not asked (no host sync in the steady state).
    if self._host_steps >= self.step_limit and bool(done.any()):
      if bool(done.all()):
        self._host_steps = 0
      obs = self.reset(done)
    return obs, reward, done

  def close(self):
    self.physics.close()


class CheetahRun(TorchBatchedEnv):
  pass


class Humanoid(TorchBatchedEnv):
  """Humanoid stand / walk / run (suite/humanoid.py:132-207) on device."""

  _MODEL = 'humanoid.xml'
  _OUTPUTS = ('xpos', 'xmat')
  _CONTROL_TIMESTEP = .025
  _STAND_HEIGHT = 1.4
  _TORSO = 'torso'
  _UPRIGHT = 8              # xmat entry of the torso that measures uprightness ('zz')
  _SIDES = ('left_', 'right_')
  _COM_VEL_SENSOR = 'torso_subtreelinvel'
  move_speed = 0.0

  def __init__(self, batch_size, move_speed=0.0, time_limit=25.0, **kw):
    self.move_speed = float(move_speed)
    super().__init__(batch_size, time_limit=time_limit, **kw)

  def _initialize_episode(self, mask):
    """randomize_limited_and_rotational_joints, redrawn while anything is in contact (suite/humanoid.py:160-165):
    every round draws the environments still pending, evaluates mj_forward for them alone and keeps those with
    ncon == 0.  One host sync per round (the loop's exit test), at reset time only."""
    torch = self.torch
    pending = mask.clone()
    for rounds in range(1, 10001):
      self.qpos.copy_(torch.where(pending[None, :], self._q0, self.qpos))
      self._randomize(pending, BatchedPhysics.RAND_ALL)
      for t in (self.qvel, self.ctrl, self.warm):
        self._zero_rows(t, pending)
      self._launch_only(pending, lambda: self.physics.forward(disable_actuation=True, stream=self._stream()))
      pending = pending & (self.ncon[0] > 0)
      if not bool(pending.any()):
        break
    else:
      raise RuntimeError('could not find collision-free start states')
    self.reset_rounds = rounds

  # rows of the (3*nbody, B) / (9*nbody, B) arrays
  def _xpos(self, body):
    b = self._body(body)
    return self.xpos[3*b:3*b + 3]

  def _xmat(self, body):
    b = self._body(body)
    return self.xmat[9*b:9*b + 9]

  def head_height(self):
    return self._xpos('head')[2]

  def torso_upright(self):
    return self._xmat(self._TORSO)[self._UPRIGHT]

  def center_of_mass_velocity(self):
    adr = self.model.sensor_adr[self.model.name2id(self._COM_VEL_SENSOR, 'sensor')]
    return self.sensordata[adr:adr + 3]

  def extremities(self):
    """(12, B): hands and feet relative to the torso, in the torso frame."""
    R = self._xmat(self._TORSO).reshape(3, 3, self.B)      # R[i, j] = xmat[3 i + j]
    torso = self._xpos(self._TORSO)
    out = []
    for side in self._SIDES:
      for limb in ('hand', 'foot'):
        d = self._xpos(side + limb) - torso
        out.append((d[:, None, :] * R).sum(dim=0))      # d . frame  (row vector times matrix)
    return self.torch.cat(out, dim=0)

  def observation(self):
    """(B, 67): joint_angles 21, head_height 1, extremities 12, torso_vertical 3, com_velocity 3,
    velocity 27 -- the order of Humanoid.get_observation flattened."""
    parts = [self.qpos[7:], self.head_height()[None], self.extremities(), self._xmat(self._TORSO)[6:9],
             self.center_of_mass_velocity(), self.qvel]
    return self.torch.cat(parts, dim=0).T

  def reward(self):
    torch = self.torch
    standing = tolerance(torch, self.head_height(), bounds=(self._STAND_HEIGHT, float('inf')),
                         margin=self._STAND_HEIGHT / 4)
    upright = tolerance(torch, self.torso_upright(), bounds=(0.9, float('inf')), sigmoid='linear', margin=1.9,
                        value_at_margin=0)
    stand_reward = standing * upright
    small_control = tolerance(torch, self.ctrl, margin=1, value_at_margin=0, sigmoid='quadratic').mean(dim=0)
    small_control = (4 + small_control) / 5
    horizontal = self.center_of_mass_velocity()[0:2]
    if self.move_speed == 0:
      dont_move = tolerance(torch, horizontal, margin=2).mean(dim=0)
      return small_control * stand_reward * dont_move
    speed = torch.linalg.norm(horizontal, dim=0)
    move = tolerance(torch, speed, bounds=(self.move_speed, float('inf')), margin=self.move_speed,
                     value_at_margin=0, sigmoid='linear')
    return small_control * stand_reward * (5 * move + 1) / 6


class HumanoidCMU(Humanoid):
  """Humanoid_CMU stand / walk / run (suite/humanoid_CMU.py:132-190) on device: the same task as the
  humanoid with the thorax as torso, uprightness = thorax y axis on world z, 137 observations."""

  _MODEL = 'humanoid_CMU.xml'
  _CONTROL_TIMESTEP = .02
  _TORSO = 'thorax'
  _UPRIGHT = 7              # 'zy'
  _SIDES = ('l', 'r')
  _COM_VEL_SENSOR = 'thorax_subtreelinvel'

  def __init__(self, batch_size, move_speed=0.0, time_limit=20.0, **kw):
    super().__init__(batch_size, move_speed=move_speed, time_limit=time_limit, **kw)


class _RandomJointStart(TorchBatchedEnv):
  """Start states = randomize_limited_and_rotational_joints (suite/utils/randomizers.py:35-88): limited
  hinges / sliders ~ U(range), unlimited hinges ~ U(-pi, pi), drawn on the device for every episode."""

  def _initialize_episode(self, mask):
    self.qpos.copy_(self.torch.where(mask[None, :], self._q0, self.qpos))
    self._randomize(mask, BatchedPhysics.RAND_ALL)
    for t in (self.qvel, self.ctrl, self.warm):
      self._zero_rows(t, mask)
    self.reset_rounds = 1

  def _sensor(self, name, n):
    adr = self.model.sensor_adr[self.model.name2id(name, 'sensor')]
    return self.sensordata[adr:adr + n]


class Walker(_RandomJointStart):
  """Planar walker stand / walk / run (suite/walker.py:41-132) on device."""

  _MODEL = 'walker.xml'
  _OUTPUTS = ('xpos', 'xmat')
  _CONTROL_TIMESTEP = .025
  _STAND_HEIGHT = 1.2

  def __init__(self, batch_size, move_speed=0.0, time_limit=25.0, **kw):
    self.move_speed = float(move_speed)
    super().__init__(batch_size, time_limit=time_limit, **kw)

  def torso_height(self):
    b = self._body('torso')
    return self.xpos[3*b + 2]

  def torso_upright(self):
    return self.xmat[9*self._body('torso') + 8]

  def observation(self):
    """(B, 2 (nbody-1) + 1 + nv): orientations (xx, xz of every body), height, velocity."""
    nb = self.model.nbody
    xm = self.xmat.reshape(nb, 9, self.B)[1:]
    orient = self.torch.stack([xm[:, 0], xm[:, 2]], dim=1).reshape(2 * (nb - 1), self.B)
    return self.torch.cat([orient, self.torso_height()[None], self.qvel], dim=0).T

  def reward(self):
    torch = self.torch
    standing = tolerance(torch, self.torso_height(), bounds=(self._STAND_HEIGHT, float('inf')), margin=self._STAND_HEIGHT / 2)
    upright = (1 + self.torso_upright()) / 2
    stand_reward = (3 * standing + upright) / 4
    if self.move_speed == 0:
      return stand_reward
    speed = self._sensor('torso_subtreelinvel', 3)[0]
    move = tolerance(torch, speed, bounds=(self.move_speed, float('inf')), margin=self.move_speed / 2,
                     value_at_margin=0.5, sigmoid='linear')
    return stand_reward * (5 * move + 1) / 6


class Hopper(_RandomJointStart):
  """Hopper stand / hop (suite/hopper.py:62-140) on device."""

  _MODEL = 'hopper.xml'
  _OUTPUTS = ('xipos',)
  _CONTROL_TIMESTEP = .02
  _STAND_HEIGHT = 0.6
  _HOP_SPEED = 2.0

  def __init__(self, batch_size, hopping=False, time_limit=20.0, **kw):
    self.hopping = bool(hopping)
    super().__init__(batch_size, time_limit=time_limit, **kw)

  def height(self):
    return self.xipos[3*self._body('torso') + 2] - self.xipos[3*self._body('foot') + 2]

  def touch(self):
    return self.torch.log1p(self.torch.cat([self._sensor('touch_toe', 1), self._sensor('touch_heel', 1)], dim=0))

  def observation(self):
    """(B, nq-1 + nv + 2): position without the root x, velocity, touch."""
    return self.torch.cat([self.qpos[1:], self.qvel, self.touch()], dim=0).T

  def reward(self):
    torch = self.torch
    standing = tolerance(torch, self.height(), bounds=(self._STAND_HEIGHT, 2))
    if self.hopping:
      speed = self._sensor('torso_subtreelinvel', 3)[0]
      hopping = tolerance(torch, speed, bounds=(self._HOP_SPEED, float('inf')), margin=self._HOP_SPEED / 2,
                          value_at_margin=0.5, sigmoid='linear')
      return standing * hopping
    small_control = tolerance(torch, self.ctrl, margin=1, value_at_margin=0, sigmoid='quadratic').mean(dim=0)
    return standing * (small_control + 4) / 5


class Quadruped(TorchBatchedEnv):
  """Quadruped walk / run (suite/quadruped.py:281-342) on device.  Start states: a random orientation,
  raised in 1 cm steps from the floor until nothing touches (`_find_non_contacting_height`), evaluated for
  all environments being reset at once."""

  _MODEL = 'quadruped.xml'
  _OUTPUTS = ('xmat',)
  _CONTROL_TIMESTEP = .02

  def __init__(self, batch_size, desired_speed=0.5, time_limit=20.0, **kw):
    self.desired_speed = float(desired_speed)
    super().__init__(batch_size, time_limit=time_limit, **kw)

  def _model_xml(self):
    from dm_control_amd.suite import quadruped
    return quadruped.make_model(floor_size=20 * self.desired_speed)

  def _initialize_episode(self, mask):
    """A random orientation (uniform on the sphere, suite/quadruped.py:243-246), raised from the floor in 1 cm steps
    until nothing touches (`_find_non_contacting_height`), for the masked environments at once."""
    torch = self.torch
    self.qpos.copy_(torch.where(mask[None, :], self._q0, self.qpos))
    self._randomize(mask, BatchedPhysics.RAND_QUATERNION | BatchedPhysics.RAND_FREE_NORMAL)
    self.qpos[2].copy_(torch.where(mask, torch.zeros_like(self.qpos[2]), self.qpos[2]))
    for t in (self.qvel, self.ctrl, self.warm):
      self._zero_rows(t, mask)
    pending = mask.clone()
    for rounds in range(1, 10001):
      self._launch_only(pending, lambda: self.physics.forward(disable_actuation=True, stream=self._stream()))
      pending = pending & (self.ncon[0] > 0)
      if not bool(pending.any()):
        break
      self.qpos[2] += 0.01 * pending.to(self.dtype)
    else:
      raise RuntimeError('Failed to find a non-contacting configuration.')
    self.reset_rounds = rounds

  def _setup(self):
    torch, m = self.torch, self.model
    hinge = [j for j in range(m.njnt) if m.jnt_type[j] == 3]
    self._hq = torch.tensor([m.jnt_qposadr[j] for j in hinge], device=self.device)
    self._hv = torch.tensor([m.jnt_dofadr[j] for j in hinge], device=self.device)
    rows = lambda types: torch.tensor([m.sensor_adr[i] + k for i in range(m.nsensor) if m.sensor_type[i] in types
                                       for k in range(3)], device=self.device)
    self._imu_rows = rows((1, 3))          # accelerometer, gyro -- in sensor order, as Physics.imu
    self._ft_rows = rows((4, 5))           # force, torque
    adr = m.sensor_adr[m.name2id('velocimeter', 'sensor')]
    self._vel_rows = slice(adr, adr + 3)
    self._torso = self._body('torso')

  def torso_upright(self):
    return self.xmat[9*self._torso + 8]

  def observation(self):
    """(B, 78): egocentric_state 44 (hinge qpos, hinge qvel, act), torso_velocity 3, torso_upright 1,
    imu 6, force_torque 24 (arcsinh) -- the order of `_common_observations`."""
    torch = self.torch
    parts = [self.qpos[self._hq], self.qvel[self._hv], self.act, self.sensordata[self._vel_rows],
             self.torso_upright()[None], self.sensordata[self._imu_rows], torch.asinh(self.sensordata[self._ft_rows])]
    return torch.cat(parts, dim=0).T

  def reward(self):
    torch = self.torch
    upright = tolerance(torch, self.torso_upright(), bounds=(1.0, float('inf')), sigmoid='linear', margin=2.0,
                        value_at_margin=0)
    move = tolerance(torch, self.sensordata[self._vel_rows][0], bounds=(self.desired_speed, float('inf')),
                     margin=self.desired_speed, value_at_margin=0.5, sigmoid='linear')
    return upright * move


_TASKS = {
    ('cheetah', 'run'): (CheetahRun, {}),
    ('walker', 'stand'): (Walker, dict(move_speed=0)),
    ('walker', 'walk'): (Walker, dict(move_speed=1)),
    ('walker', 'run'): (Walker, dict(move_speed=8)),
    ('hopper', 'stand'): (Hopper, dict(hopping=False)),
    ('hopper', 'hop'): (Hopper, dict(hopping=True)),
    ('humanoid', 'stand'): (Humanoid, dict(move_speed=0)),
    ('humanoid', 'walk'): (Humanoid, dict(move_speed=1)),
    ('humanoid', 'run'): (Humanoid, dict(move_speed=10)),
    ('quadruped', 'walk'): (Quadruped, dict(desired_speed=0.5)),
    ('quadruped', 'run'): (Quadruped, dict(desired_speed=5)),
    ('humanoid_CMU', 'stand'): (HumanoidCMU, dict(move_speed=0)),
    ('humanoid_CMU', 'walk'): (HumanoidCMU, dict(move_speed=1)),
    ('humanoid_CMU', 'run'): (HumanoidCMU, dict(move_speed=10)),
}


def make(domain, task, batch_size, **kwargs):
  """Device-resident batched version of `suite.load(domain, task)`."""
  if (domain, task) not in _TASKS:
    raise ValueError('no device-resident task for (%r, %r); available: %s' % (domain, task, sorted(_TASKS)))
  cls, kw = _TASKS[(domain, task)]
  kw = dict(kw)
  kw.update(kwargs)
  return cls(batch_size, **kw)
