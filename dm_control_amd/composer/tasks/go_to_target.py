"""BASELINE config 4 as an environment: the CMU humanoid (2019, position-controlled) going to a target on the Floor.

Batch / device version of `locomotion.tasks.go_to_target.GoToTarget` (dm_control/locomotion/tasks/go_to_target.py)
with `walkers.CMUHumanoidPositionControlled` (walkers/cmu_humanoid.py:360-399) on `arenas.Floor` (arenas/floors.py),
the composition of `locomotion/examples/basic_cmu_2019.py:97-118`.  The physics is the restated asset
`suite/assets/cmu_2019_position_floor.xml` (see scripts/make_cmu_floor_model.py); this module is the task layer:

  * `initialize_episode` (go_to_target.py:139-165): walker at its upright pose (the asset's qpos0,
    cmu_humanoid.py:174-176) shifted to a uniform position of the arena, optional yaw; target at a uniform position;
  * `before_step` -> `walker.apply_action` (`physics.bind(actuators).ctrl = action`);
  * `after_step` (:187-211): failure = any contact between a non-foot walker geom and the ground (the contact scan
    over `physics.data.contact`), moving-target resampling after `steps_before_moving_target` rewarded steps;
  * reward 1 inside `distance_tolerance` of the target, discount 0 on failure, termination on failure;
  * observations: the walker's proprioception, kinematic and dynamic sensors, touch (:108-118; observable
    definitions legacy_base.py:196-330, cmu_humanoid.py:436-499) and the egocentric target vector.

The target is a massless site in the reference (moved by editing the MJCF and recompiling); it has no physics, so
here it is a per-environment (2, B) tensor of the task -- no per-env model delta is involved.
"""
import numpy as np

from dm_control_amd import mjcf_compiler, observation
from dm_control_amd.composer import environment
from dm_control_amd.composer.physics import DevicePhysics
from dm_control_amd.suite import common

DEFAULT_DISTANCE_TOLERANCE_TO_TARGET = 1.0      # go_to_target.py:25
_TOUCH_THRESHOLD = 1e-3                         # legacy_base.py:28
_TORQUE_THRESHOLD = 60                          # cmu_humanoid.py:181
_ASSET = 'cmu_2019_position_floor'


class CMUHumanoidWalker(environment.Entity):
  """What the task needs of `CMUHumanoidPositionControlled`: element groups resolved against the compiled model."""

  root_body = 'root'
  head = 'head'
  end_effectors = ('rradius', 'lradius', 'rfoot', 'lfoot')          # cmu_humanoid.py:304-309
  _foot_bodies = ('lfoot', 'ltoes', 'rfoot', 'rtoes')                 # lfoot / rfoot and their descendants (:298-301)

  def __init__(self, model):
    m = self.model = model
    self.actuators = list(m.names['actuator'])
    # observable_joints: the joint of every actuator, in actuator order (cmu_humanoid.py:311-314)
    self.observable_joints = [m.names['joint'][int(j)] for j in np.asarray(m.actuator_trnid)[:, 0]]
    foot_ids = {m.name2id(b, 'body') for b in self._foot_bodies}
    walker_bodies = set(range(m.name2id('root', 'body'), m.nbody))
    self.ground_contact_geoms = [n for i, n in enumerate(m.names['geom']) if int(m.geom_bodyid[i]) in foot_ids]
    self.nonfoot_geoms = [n for i, n in enumerate(m.names['geom'])
                          if int(m.geom_bodyid[i]) in walker_bodies and int(m.geom_bodyid[i]) not in foot_ids]
    self.touch_sensors = [n for i, n in enumerate(m.names['sensor']) if int(m.sensor_type[i]) == 0]
    self.torque_sensors = [n for i, n in enumerate(m.names['sensor']) if int(m.sensor_type[i]) == 5]
    self.prev_action = None

  def apply_action(self, physics, action, random_state):
    """legacy_base / cmu_humanoid apply_action: the action IS the control of the scaled position actuators."""
    del random_state
    self.prev_action = action
    physics.field('ctrl').copy_(action.T.to(physics.dtype))


class GoToTarget(environment.Task):
  # the reference's observation update runs mj_forward first (no substep hook reads through a binding before it): see
  # composer.Environment.step
  observation_forward = True

  def __init__(self, model=None, moving_target=False, target_relative=False, target_relative_dist=1.5,
               steps_before_moving_target=10, distance_tolerance=DEFAULT_DISTANCE_TOLERANCE_TO_TARGET,
               walker_spawn_rotation=None, arena_size=(8.0, 8.0), physics_timestep=0.005, control_timestep=0.03):
    self.model = model or mjcf_compiler.compile_xml(common.read_model(_ASSET + '.xml'))
    self.walker = CMUHumanoidWalker(self.model)
    self._arena_half = np.asarray(arena_size, dtype=np.float64) / 2      # arena_position: U(-size/2, size/2)
    self._distance_tolerance = distance_tolerance
    self._moving_target = moving_target
    self._target_relative = target_relative
    self._target_relative_dist = target_relative_dist
    self._steps_before_moving_target = steps_before_moving_target
    self._walker_spawn_rotation = walker_spawn_rotation
    self.set_timesteps(control_timestep=control_timestep, physics_timestep=physics_timestep)
    m = self.model
    self._root = m.name2id(self.walker.root_body, 'body')
    ground = {m.name2id('groundplane', 'geom')}
    ground.add(0)      # go_to_target.py:165 adds the TARGET SITE's element id (0) to the ground geom ids
    self._ground_ids = sorted(ground)
    self._nonfoot_ids = [m.name2id(n, 'geom') for n in self.walker.nonfoot_geoms]
    w = self.walker
    root, jn = w.root_body, w.observable_joints
    # MJCFFeature observables, in the order the task enables them (go_to_target.py:108-112)
    self.feature_table = observation.GatherTable(m, [
        ('qpos', jn), ('qvel', jn),                                          # joints_pos, joints_vel
        ('xpos', [root], 'z'),                                               # body_height
        ('xmat', [root], ['zx', 'zy', 'zz']),                                # world_zaxis
        ('sensordata', ['sensor_root_gyro']), ('sensordata', ['sensor_root_veloc']),
        ('sensordata', ['sensor_root_accel']),                               # kinematic_sensors
        ('sensordata', w.torque_sensors, None, ('tanh2', _TORQUE_THRESHOLD)),  # sensors_torque (cmu_humanoid.py:462)
        ('sensordata', w.touch_sensors, None, ('greater', _TOUCH_THRESHOLD)),  # sensors_touch (legacy_base.py:262)
    ])
    self.feature_names = ['joints_pos', 'joints_vel', 'body_height', 'world_zaxis', 'sensors_gyro', 'sensors_velocimeter',
                          'sensors_accelerometer', 'sensors_torque', 'sensors_touch']

  @property
  def entities(self):
    return (self.walker,)

  def make_physics(self, batch_size, device_id=0, precision=32):
    caps = dict(common.DEFAULT_CAPS.get(_ASSET, {}))
    if caps.pop('precision', precision) != precision:
      raise ValueError('%s runs in fp32 only' % _ASSET)
    return DevicePhysics(self.model, batch_size, device_id=device_id, precision=precision,
                         outputs=('sensordata', 'xpos', 'xmat', 'contact_geom1'), **caps)

  # -- helpers -----------------------------------------------------------------------------------------
  def _uniform(self, physics, lo, hi, n):
    """(n, B) device samples; the host RandomState seeds a device generator once per task."""
    torch = physics.torch
    if getattr(self, '_gen', None) is None:
      self._gen = torch.Generator(device=physics.device)
      self._gen.manual_seed(int(self._seed))
    u = torch.rand((n, physics.B), generator=self._gen, device=physics.device, dtype=physics.dtype)
    lo, hi = physics.const(lo).reshape(-1, 1), physics.const(hi).reshape(-1, 1)
    return lo + (hi - lo) * u

  def generators(self):
    """Device RNG streams of the task (registered with a capturing HIP graph)."""
    return [self._gen] if getattr(self, '_gen', None) is not None else []

  def target_position(self, physics):
    return self._target

  # -- hooks -------------------------------------------------------------------------------------------
  def initialize_episode(self, physics, random_state, mask):
    torch = physics.torch
    if getattr(self, '_seed', None) is None:
      self._seed = random_state.randint(2**31 - 1)
      self._gen = torch.Generator(device=physics.device)
      self._gen.manual_seed(int(self._seed))
      self._target = torch.zeros((2, physics.B), dtype=physics.dtype, device=physics.device)
      self._failure = torch.zeros(physics.B, dtype=torch.bool, device=physics.device)
      self._reward_steps = torch.zeros(physics.B, dtype=torch.int32, device=physics.device)
      lut = torch.zeros(self.model.ngeom + 1, dtype=torch.bool, device=physics.device)
      self._is_ground, self._is_nonfoot = lut.clone(), lut.clone()
      self._is_ground[torch.tensor(self._ground_ids, device=physics.device)] = True
      self._is_nonfoot[torch.tensor(self._nonfoot_ids, device=physics.device)] = True
    m2 = mask[None, :]
    # initialize_episode_mjcf: target ~ U(arena); initialize_episode: reinitialize_pose (qpos0 of the asset is the
    # upright pose, already restored by the reset) + shift_pose to a uniform arena position (+ optional yaw)
    tgt = self._uniform(physics, -self._arena_half, self._arena_half, 2)
    self._target.copy_(torch.where(m2, tgt, self._target))
    spawn = self._uniform(physics, -self._arena_half, self._arena_half, 2)
    q = physics.field('qpos')
    q[0:2] = torch.where(m2, q[0:2] + spawn, q[0:2])
    if self._walker_spawn_rotation:
      lo, hi = self._walker_spawn_rotation
      yaw = self._uniform(physics, [lo], [hi], 1)[0]
      rot = torch.stack([torch.cos(yaw / 2), torch.zeros_like(yaw), torch.zeros_like(yaw), torch.sin(yaw / 2)])
      cur = q[3:7].clone()
      new = torch.stack([rot[0]*cur[0] - rot[3]*cur[3], rot[0]*cur[1] - rot[3]*cur[2],
                         rot[0]*cur[2] + rot[3]*cur[1], rot[0]*cur[3] + rot[3]*cur[0]])      # rot (x) cur, rot = (w,0,0,z)
      q[3:7] = torch.where(m2, new, cur)
    physics.mark_as_dirty()
    self._failure.copy_(self._failure & ~mask)
    self._reward_steps.copy_(torch.where(mask, torch.zeros_like(self._reward_steps), self._reward_steps))

  def before_step(self, physics, action, random_state):
    self.walker.apply_action(physics, action, random_state)

  def after_step(self, physics, random_state):
    torch = physics.torch
    # the contact scan of go_to_target.py:189-193 for the whole batch: contact slots hold geom ids, -1 when empty
    g1 = physics.field('contact_geom1').long()
    g2 = physics.field('contact_geom2').long()
    ng = self.model.ngeom
    g1, g2 = torch.where(g1 < 0, ng, g1), torch.where(g2 < 0, ng, g2)
    bad = (self._is_nonfoot[g1] & self._is_ground[g2]) | (self._is_ground[g1] & self._is_nonfoot[g2])
    self._failure.copy_(bad.any(dim=0))
    if self._moving_target:
      move = self._reward_steps >= self._steps_before_moving_target
      if self._target_relative:
        d = self._target_relative_dist
        new = physics.field('xpos')[3*self._root:3*self._root + 2] + self._uniform(physics, [-d, -d], [d, d], 2)
      else:
        new = self._uniform(physics, -self._arena_half, self._arena_half, 2)
      self._target.copy_(torch.where(move[None, :], new, self._target))
      self._reward_steps.copy_(torch.where(move, torch.zeros_like(self._reward_steps), self._reward_steps))

  def should_terminate_episode(self, physics):
    return self._failure

  def get_discount(self, physics):
    return (~self._failure).to(physics.dtype)

  def get_reward(self, physics):
    torch = physics.torch
    root = physics.field('xpos')[3*self._root:3*self._root + 2]
    distance = torch.linalg.norm(self._target - root, dim=0)
    hit = distance < self._distance_tolerance
    if self._moving_target:
      restarting = getattr(self, 'restarting', None)      # environments re-initialised in this call: no reward step
      counted = hit if restarting is None else hit & ~restarting
      self._reward_steps += counted.to(torch.int32)
    return hit.to(physics.dtype)

  # -- observations --------------------------------------------------------------------------------------
  def egocentric(self, physics, vec3):
    """walker.transform_vec_to_egocentric_frame: world vector (3, B) in the root body's frame (v . xmat)."""
    R = physics.field('xmat')[9*self._root:9*self._root + 9].reshape(3, 3, physics.B)
    return (vec3[:, None, :] * R).sum(dim=0)

  def get_observation(self, physics):
    """dict of (B, n) tensors, the enabled observables of go_to_target.py:108-118."""
    torch = physics.torch
    flat = physics.gather(self.feature_table)                       # ONE gather launch for every MJCFFeature
    obs = {}
    for name, (_, first, count) in zip(self.feature_names, self.feature_table.slices):
      obs[name] = flat[:, first:first + count]
    # joints_pos / joints_vel were two entries each span (qpos, qvel): the table keeps them in order
    xpos = physics.field('xpos')
    root = xpos[3*self._root:3*self._root + 3]
    m = self.model
    eff = [m.name2id(b, 'body') for b in self.walker.end_effectors]
    rel = torch.stack([xpos[3*b:3*b + 3] - root for b in eff])                       # (4, 3, B)
    R = physics.field('xmat')[9*self._root:9*self._root + 9].reshape(3, 3, physics.B)
    ego = (rel[:, :, None, :] * R[None]).sum(dim=1)                                  # (4, 3, B)
    obs['end_effectors_pos'] = ego.reshape(12, physics.B).T
    head = xpos[3*m.name2id(self.walker.head, 'body'):][:3] - root
    obs['appendages_pos'] = torch.cat([ego.reshape(12, physics.B), self.egocentric(physics, head)], dim=0).T
    tgt3 = torch.cat([self._target, torch.zeros_like(self._target[:1])], dim=0)
    obs['target'] = self.egocentric(physics, tgt3 - root).T
    return obs


def make(batch_size, device_id=0, precision=32, time_limit=30.0, random_state=0, observation_options=None,
         delayed_observation_padding='zero', strip_singleton_obs_buffer_dim=True, **task_kwargs):
  """`composer.Environment(GoToTarget(CMUHumanoidPositionControlled(), Floor()), time_limit=30)` for a batch
  (locomotion/examples/basic_cmu_2019.py:97-118: physics 0.005 s, control 0.03 s).  `observation_options` & co:
  composer/updater.py (buffered / delayed / aggregated observables)."""
  task = GoToTarget(**task_kwargs)
  physics = task.make_physics(batch_size, device_id=device_id, precision=precision)
  return environment.Environment(task, physics, time_limit=time_limit, random_state=random_state,
                                 observation_options=observation_options, delayed_observation_padding=delayed_observation_padding,
                                 strip_singleton_obs_buffer_dim=strip_singleton_obs_buffer_dim)
