"""BASELINE config 5 as an environment: locomotion.soccer 2-vs-2 with BoxHead walkers, for a batch on device.

Task layer of `dm_control.locomotion.soccer.load(team_size=2, walker_type=BOXHEAD)` (soccer/__init__.py:92-148,
soccer/task.py:36-230) over `suite/assets/soccer_2v2_boxhead.xml`, which is the XML the reference's own PyMJCF
composition produces for that call (scripts/make_pymjcf_goldens.py; element names are PyMJCF's: `home0/root_x/`,
`soccer_ball/`, `//unnamed_geom_1..4` for the four walls):

  * 4 agents: the action is (B, 4, 3) = per player (roll, steer, kick), written to the players' actuators
    (`walker.apply_action`, task.py:211-213);
  * `UniformInitializer` (soccer/initializers.py:33-127): ball and players uniform over `spawn_ratio` of the pitch,
    players with a uniform yaw; placements with two entities closer than their bounding radii are redrawn (the
    reference redraws on detected contacts);
  * goal / out-of-court detection by `PositionDetector` volumes on the ball position (entities/props/
    position_detector.py; pitch.py:426-456), evaluated AFTER EVERY SUBSTEP and retained until the end of the control
    step (`retain_substep_detections=True`, pitch.py:262): the detectors are entities with an `after_substep`
    hook.  They only look at the ball's position, which the step kernel records after every physics step of a launch
    (the substep probe, dmc_batch_set_step_probe): the control step is ONE launch and the detections are the
    per-substep ones (`Environment(..., fuse_substeps=False)` runs n_sub_steps launches with the hooks in between, the
    reference's literal order, to the same result; `fuse_substeps=True` checks at the end of the control step only);
  * reward +1 / -1 per player when a team scores, discount 0 and termination on a goal (task.py:160-209);
    throw-in when the ball left the court (task.py:215-217, :128-135);
  * observations per player (soccer/observables.py CoreObservablesAdder): proprioception (kick joint position /
    velocity, body height, end effector, world z axis, prev_action), kinematic sensors, ball / teammate / opponent
    positions and velocities and goal / field corners in the player's egocentric frame, and the stats_* scalars.
    Observation tensors are (B, 4, n).

`randomize_pitch=(min_size, max_size)` is `RandomizedPitch` (pitch.py:604-669; `soccer.load` uses (32, 24) ..
(48, 36), soccer/__init__.py:140-148): every episode draws its own pitch size; walls, goal posts (positions,
lengths, radii), goal / field detectors and spawn ranges follow per environment.  The reference edits the MJCF and
recompiles; a batch shares one compiled model, so the 4 walls and 20 goal posts are declared per-environment geoms
(`DevicePhysics.declare_env_geoms`, dmc_batch_set_env_geoms) and their rows of the 'env_geom' tensor are rewritten on
device under the reset mask.  Default: the fixed 40 x 30 pitch of the asset.
"""
import ctypes
import os

import numpy as np

from dm_control_amd import mjcf_compiler
from dm_control_amd.composer import environment
from dm_control_amd.composer.physics import DevicePhysics
from dm_control_amd.suite import common

_ASSET = 'soccer_2v2_boxhead'
_PLAYERS = ('home0', 'home1', 'away0', 'away1')
_TEAM = (0, 0, 1, 1)                                   # 0 = HOME (defends -x), 1 = AWAY
_SPOTS = np.zeros((4, 2))      # attachment frames of the players: PyMJCF attaches every walker at the origin
_SIZE = np.array([40.0, 30.0])
_SIDE_WIDTH = 32. / 6.
_GOAL_SIZE = np.array([_SIDE_WIDTH / 2, _SIZE[1] * 0.33, _SIDE_WIDTH / 2])       # pitch.py _get_goal_size
_INIT_BALL_Z, _SPAWN_RATIO, _THROW_IN_BALL_Z = 0.5, 0.6, 0.5
_GOALPOST_RELATIVE_SIZE, _SUPPORT_POST_RATIO = 0.07, 0.75                          # pitch.py:40-41
# unit from / to of the goal posts in goal coordinates (pitch.py:165-230), the table scripts/make_soccer_model.py used
_GOALPOSTS = {'right_post': (1, -1, -1, 1, -1, 1), 'left_post': (1, 1, -1, 1, 1, 1), 'top_post': (1, -1, 1, 1, 1, 1),
              'right_base': (1, -1, -1, -1, -1, -1), 'left_base': (1, 1, -1, -1, 1, -1), 'back_base': (-1, -1, -1, -1, 1, -1),
              'right_support': (-1, -1, -1, .2, -1, 1), 'right_top_support': (.2, -1, 1, 1, -1, 1),
              'left_support': (-1, 1, -1, .2, 1, 1), 'left_top_support': (.2, 1, 1, 1, 1, 1)}


def addresses(model):
  """qpos / qvel addresses of the composed model by PyMJCF element name: {'players': [(qx, qy) x 4 in _PLAYERS order],
  'ball_q': first qpos of the ball's free joint, 'ball_v': first dof} -- for callers that place entities directly
  (bench.py, tests) instead of through the task's initializer."""
  jq = lambda n: int(model.jnt_qposadr[model.name2id(n, 'joint')])
  return dict(players=[(jq(p + '/root_x/'), jq(p + '/root_y/')) for p in _PLAYERS], ball_q=jq('soccer_ball/'),
              ball_v=int(model.jnt_dofadr[model.name2id('soccer_ball/', 'joint')]))


KICKOFF_SPOTS = np.array([(-10., 5.), (-10., -5.), (10., 5.), (10., -5.)])      # where bench / tests stand the players


def kickoff_qpos(model):
  """qpos0 with the players on KICKOFF_SPOTS and the ball above the centre spot at the initializer's height
  (soccer/initializers.py:31 `_INIT_BALL_Z`): PyMJCF's own qpos0 has all five entities at the origin."""
  q = np.array(model.qpos0, dtype=float)
  a = addresses(model)
  for (qx, qy), xy in zip(a['players'], KICKOFF_SPOTS):
    q[qx], q[qy] = xy
  q[a['ball_q'] + 2] = 0.5
  return q



# ---- the task layer as two kernels (soccer_task.hip) ----------------------------------------------------------------
def _args_struct(real):
  P = ctypes.c_void_p
  ints = [(n, ctypes.c_int) for n in ('B', 'nq', 'nv', 'nu', 'nsub', 'rounds', 'ball_q', 'ball_v', 'ball_geom', 'ball_linvel')]
  fields = ints + [('place_rows', ctypes.c_int * 15), ('ctrl_rows', ctypes.c_int * 12), ('root', ctypes.c_int * 4), ('ncomp', ctypes.c_int),
                   ('time_limit', ctypes.c_double), ('spawn_ratio', real), ('ball_z', real), ('pi', real)]
  fields += [(n, P) for n in ('qpos', 'qvel', 'ctrl', 'warm', 'time', 'env_mode', 'warning', 'qpos0', 'size', 'state', 'lo', 'hi',
                              'prev_action', 'geom_xpos', 'xpos', 'xmat', 'cvel', 'sensordata', 'action', 'u_place', 'u_shrink',
                              'trace', 'reset_next', 'reward', 'discount', 'step_type', 'comp')]

  class SoccerArgs(ctypes.Structure):
    _fields_ = fields
  return SoccerArgs


_KERNELS = {}


def task_kernels(precision, verbose=False):
  """(library, SoccerArgs) of soccer_task.hip for fp32 / fp64: compiled by hipcc for gfx950 on first use into the in-tree
  plugin cache (as the model-specialised step kernels and the suite's generated task layers are)."""
  k = _KERNELS.get(precision)
  if k is None:
    from dm_control_amd.suite import fused_env
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'soccer_task.hip')) as f:
      src = '#define DMC_SOCCER_REAL %s\n' % ('float' if precision == 32 else 'double') + f.read()
    lib = ctypes.CDLL(fused_env._compile(src, verbose=verbose))
    S = _args_struct(ctypes.c_float if precision == 32 else ctypes.c_double)
    if lib.soccer_args_size() != ctypes.sizeof(S):
      raise RuntimeError('soccer_task.hip: SoccerArgs is %d bytes on the device side, %d here' % (lib.soccer_args_size(), ctypes.sizeof(S)))
    for fn in (lib.soccer_pre, lib.soccer_post):
      fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(S)]
      fn.restype = ctypes.c_int
    k = _KERNELS[precision] = (lib, S)
  return k


class PositionDetector:
  """Axis-aligned detection volume on the ball's geom position (position_detector.py:42-260).  State and bounds live in
  the task's `DetectorBank` (one evaluation for all detectors); this object is the per-detector view of them."""

  def __init__(self, lower, upper, inverted=False, retain_substep_detections=False):
    self.lower, self.upper = np.asarray(lower, float), np.asarray(upper, float)
    self.mid = (self.lower + self.upper) / 2
    self.inverted = inverted
    self.retain = retain_substep_detections
    self.detected = None      # (B,) bool: a row of the bank's state
    self._lo_t = self._hi_t = None

  def bounds(self, physics):
    """(lower, upper) as (d, B) tensors: per-environment once the pitch is randomised (`resize`)."""
    self._bank.materialise(physics)
    return self._lo_t, self._hi_t

  def resize(self, physics, pos, size, mask):
    """position_detector.py:149-160 for the masked environments; pos / size: (d, B) tensors."""
    lo, hi = self.bounds(physics)
    m2 = mask[None, :]
    lo.copy_(physics.torch.where(m2, pos - size, lo))
    hi.copy_(physics.torch.where(m2, pos + size, hi))


class DetectorBank(environment.Entity):
  """The pitch's PositionDetectors as ONE entity: their hooks (initialize_episode, before_step, after_substep;
  position_detector.py:200-260) evaluated for all detectors in a handful of tensor operations instead of a handful per
  detector.  `after_substeps` is the same from the step kernel's substep probe (environment.Environment: the control
  step stays one launch and the detections are the per-substep ones)."""

  substep_probe_geom = 'soccer_ball/geom'

  def __init__(self, detectors, ball_xpos):
    self.detectors = list(detectors)
    self._ball_xpos = ball_xpos
    self.state = None
    for d in self.detectors:
      d._bank = self

  def materialise(self, physics):
    if self.state is not None:
      return
    torch = physics.torch
    n, B, dev = len(self.detectors), physics.B, physics.device
    self.state = torch.zeros((n, B), dtype=torch.bool, device=dev)
    # a 2-d detector (the field) is unbounded in z
    self.lo = torch.full((n, 3, B), -np.inf, dtype=physics.dtype, device=dev)
    self.hi = torch.full((n, 3, B), np.inf, dtype=physics.dtype, device=dev)
    for k, d in enumerate(self.detectors):
      nd = len(d.lower)
      self.lo[k, :nd] = torch.as_tensor(d.lower, dtype=physics.dtype, device=dev)[:, None]
      self.hi[k, :nd] = torch.as_tensor(d.upper, dtype=physics.dtype, device=dev)[:, None]
      d._lo_t, d._hi_t, d.detected = self.lo[k, :nd], self.hi[k, :nd], self.state[k]
    self.inverted = torch.as_tensor([d.inverted for d in self.detectors], dtype=torch.bool, device=dev)[None, :, None]
    self.retain = torch.as_tensor([d.retain for d in self.detectors], dtype=torch.bool, device=dev)[:, None]

  def initialize_episode(self, physics, random_state, mask):
    self.materialise(physics)
    self.state.masked_fill_(mask[None, :], False)

  def before_step(self, physics, random_state):
    # position_detector.py before_step: a retained detection is cleared at the start of the next control step
    self.state.masked_fill_(self.retain, False)

  def after_substep(self, physics, random_state):
    self.after_substeps(physics, self._ball_xpos(physics)[None])

  def after_substeps(self, physics, trace):
    """trace: (n_sub_steps, 3, B) ball positions.  after_substep applied n times: a retaining detector ORs its
    per-substep detections into the control step's, a plain one keeps the last."""
    torch = physics.torch
    p = trace[:, None]                                                           # (n, 1, 3, B)
    inside = ((p > self.lo[None]) & (p < self.hi[None])).all(dim=2)             # (n, detectors, B)
    now = inside ^ self.inverted
    self.state.copy_(torch.where(self.retain, self.state | now.any(dim=0), now[-1]))


class Soccer2v2(environment.Task):

  def __init__(self, model=None, control_timestep=0.025, spawn_ratio=_SPAWN_RATIO, randomize_pitch=None, task_kernels=True):
    self._task_kernels = bool(task_kernels)      # device_step: the control step's task layer as two kernels (soccer_task.hip)
    self.model = model or mjcf_compiler.compile_xml(common.read_model(_ASSET + '.xml'))
    self.set_timesteps(control_timestep=control_timestep, physics_timestep=0.005)      # task.py:105-106
    self._spawn_ratio = spawn_ratio
    self._randomize = None if randomize_pitch is None else (np.asarray(randomize_pitch[0], float), np.asarray(randomize_pitch[1], float))
    self._size_t = None                      # (2, B) pitch half-sizes per environment
    gs = _GOAL_SIZE
    home_pos = np.array([-_SIZE[0] + gs[0], 0, gs[2]])
    away_pos = np.array([_SIZE[0] - gs[0], 0, gs[2]])
    self.home_goal = PositionDetector(home_pos - gs, home_pos + gs, retain_substep_detections=True)
    self.away_goal = PositionDetector(away_pos - gs, away_pos + gs, retain_substep_detections=True)
    fs = np.array([_SIZE[0] - 2 * gs[0], _SIZE[1] - 2 * gs[0]])
    self.field = PositionDetector(-fs, fs, inverted=True)
    self.detectors = DetectorBank((self.home_goal, self.away_goal, self.field), self.ball_xpos)
    m = self.model
    self._ball_geom = m.name2id('soccer_ball/geom', 'geom')
    self._ball_body = m.name2id('soccer_ball/', 'body')
    self._root = [m.name2id(p + '/head_body', 'body') for p in _PLAYERS]
    jq = lambda n: int(m.jnt_qposadr[m.name2id(n, 'joint')])
    jv = lambda n: int(m.jnt_dofadr[m.name2id(n, 'joint')])
    jn = lambda p, k: '%s/%s%s' % (p, k, '/' if k.startswith('root_') else '')      # attachment-frame joints: `home0/root_x/`
    self._q = {p: {k: jq(jn(p, k)) for k in ('root_x', 'root_y', 'root_z', 'steer', 'kick', 'roll')} for p in _PLAYERS}
    self._v = {p: {k: jv(jn(p, k)) for k in ('root_x', 'root_y', 'root_z', 'steer', 'kick', 'roll')} for p in _PLAYERS}
    self._ball_q, self._ball_v = jq('soccer_ball/'), jv('soccer_ball/')
    self._ctrl_rows = [[m.name2id('%s/%s' % (p, a), 'actuator') for a in ('roll', 'steer', 'kick')] for p in _PLAYERS]
    sadr = lambda n: int(m.sensor_adr[m.name2id(n, 'sensor')])
    self._sens = {p: {k: sadr('%s/sensor_torso_%s' % (p, k)) for k in ('vel', 'gyro', 'accel')} for p in _PLAYERS}
    self._sadr = {n: int(a) for n, a in zip(m.names['sensor'], m.sensor_adr) if n}      # sensor name -> sensordata row
    self._ball_linvel = sadr('soccer_ball/linear_velocity')
    self._ball_angvel = sadr('soccer_ball/angular_velocity')
    self._prev_action = None
    self._gen = None

  @property
  def entities(self):
    return (self.detectors,)

  def generators(self):
    return [self._gen] if self._gen is not None else []

  def make_physics(self, batch_size, device_id=0, precision=32):
    caps = dict(common.DEFAULT_CAPS.get(_ASSET, {}))
    caps.pop('precision', None)
    physics = DevicePhysics(self.model, batch_size, device_id=device_id, precision=precision,
                            outputs=('sensordata', 'xpos', 'xmat', 'geom_xpos', 'cvel'), **caps)
    if self._randomize is not None:
      physics.declare_env_geoms(self.pitch_geoms())
    return physics

  def pitch_geoms(self):
    """The geoms a pitch resize moves: 4 walls, then the 10 posts of each goal."""
    return ['//unnamed_geom_%d' % (k + 1) for k in range(4)] + ['%s/%s' % (g, p) for g in ('home_goal', 'away_goal') for p in _GOALPOSTS]

  def _resize_pitch(self, physics, size, mask):
    """RandomizedPitch.initialize_episode_mjcf (pitch.py:645-669) for the masked environments; size: (2, B)."""
    torch = physics.torch
    B = physics.B
    eg = physics.field('env_geom')
    m2 = mask[None, :]
    z = torch.zeros(B, dtype=physics.dtype, device=physics.device)
    one = torch.ones_like(z)
    sx, sy = size[0], size[1]
    gs = torch.stack([one * (_SIDE_WIDTH / 2), sy * 0.33, one * (_SIDE_WIDTH / 2)])           # _get_goal_size
    # walls (_wall_pos_xyaxes): orientation fixed, position follows the size
    for k, pos in enumerate((torch.stack([z, -sy, z]), torch.stack([z, sy, z]), torch.stack([-sx, z, z]), torch.stack([sx, z, z]))):
      eg[16*k:16*k + 3] = torch.where(m2, pos, eg[16*k:16*k + 3])
    radius = _GOALPOST_RELATIVE_SIZE * gs.sum(dim=0) / 3
    k = 4
    for direction, gx in ((1.0, -sx + gs[0]), (-1.0, sx - gs[0])):
      gpos = torch.stack([gx, z, gs[2]])
      d3 = physics.const([direction, direction, 1.0])[:, None]
      for pname, unit in _GOALPOSTS.items():
        u = physics.const(unit)
        frm = u[:3, None] * d3 * gs + gpos
        to = u[3:, None] * d3 * gs + gpos
        half = 0.5 * torch.linalg.norm(to - frm, dim=0)
        r = radius * (1.01 if 'top' in pname else 1.0) * (_SUPPORT_POST_RATIO if 'support' in pname else 1.0)
        rows = torch.cat([0.5 * (frm + to), eg[16*k + 3:16*k + 12], torch.stack([r, half, z]), (r + half)[None]])
        eg[16*k:16*k + 16] = torch.where(m2, rows, eg[16*k:16*k + 16])
        k += 1
    home = torch.stack([-sx + gs[0], z, gs[2]]); away = torch.stack([sx - gs[0], z, gs[2]])
    self.home_goal.resize(physics, home, gs, mask)
    self.away_goal.resize(physics, away, gs, mask)
    self.field.resize(physics, torch.zeros((2, B), dtype=physics.dtype, device=physics.device),
                      torch.stack([sx - 2 * gs[0], sy - 2 * gs[0]]), mask)

  # -- helpers -----------------------------------------------------------------------------------------
  def ball_xpos(self, physics):
    g = self._ball_geom
    return physics.field('geom_xpos')[3*g:3*g + 3]

  def _uniform(self, physics, lo, hi):
    torch = physics.torch
    lo, hi = physics.const(lo).reshape(-1, 1), physics.const(hi).reshape(-1, 1)
    u = torch.rand((lo.shape[0], physics.B), generator=self._gen, device=physics.device, dtype=physics.dtype)
    return lo + (hi - lo) * u

  def _place(self, physics, mask):
    """UniformInitializer._initialize_entities (soccer/initializers.py:96-127) for the masked environments: ball and
    players uniform over the spawn range, players with a uniform yaw; a placement with two entities closer than their
    bounding radii is redrawn (the reference redraws on detected contacts), up to four times.  All four candidate
    draws are made at once and every environment takes its first acceptable one: no data-dependent loop, one random
    tensor, one scatter into qpos (the per-entity version was ~150 small operations in EVERY control step, because the
    restart mask is only known on the device)."""
    torch = physics.torch
    B = physics.B
    q = physics.field('qpos')
    R = 4
    u = torch.rand((R, 5, 3, B), generator=self._gen, device=physics.device, dtype=physics.dtype) * 2 - 1
    spawn = self._size_t * self._spawn_ratio                                     # (2, B): arena.size * spawn_ratio
    xy = u[:, :, :2] * spawn[None, None]                                         # (R, 5, 2, B): ball, then the players
    d = torch.linalg.norm(xy[:, :, None] - xy[:, None, :], dim=3)                # (R, 5, 5, B)
    close = ((d < 1.5) & ~self._eye5).flatten(1, 2).any(dim=1)                   # (R, B)
    ok = ~close
    ok[R - 1] = True                                                             # the last draw is kept whatever it is
    first = ok.to(torch.int8).argmax(dim=0)                                      # first acceptable round per environment
    sel = torch.gather(u, 0, first[None, None, None, :].expand(1, 5, 3, B))[0]   # (5, 3, B)
    pos = sel[:, :2] * spawn[None]
    yaw = sel[1:, 2] * np.pi                                                     # (4, B)
    z = torch.full((1, B), _INIT_BALL_Z - 0.35, dtype=physics.dtype, device=physics.device)      # geom centre at z = 0.5 (body + 0.35)
    vals = torch.cat([pos[0], z, (pos[1:] - self._spots_t).reshape(8, B), yaw])  # rows: ball x y z, players x y ..., yaws
    cur = q[self._place_rows]
    q.index_copy_(0, self._place_rows, torch.where(mask[None, :], vals, cur))

  # -- hooks -------------------------------------------------------------------------------------------
  def initialize_episode(self, physics, random_state, mask):
    torch = physics.torch
    if self._gen is None:
      self._gen = torch.Generator(device=physics.device)
      self._gen.manual_seed(int(random_state.randint(2**31 - 1)))
      self._prev_action = torch.zeros((4, 3, physics.B), dtype=physics.dtype, device=physics.device)
      self._size_t = torch.as_tensor(_SIZE, dtype=physics.dtype, device=physics.device)[:, None].expand(2, physics.B).clone()
      dev = physics.device
      self._eye5 = torch.eye(5, dtype=torch.bool, device=dev)[None, :, :, None]
      self._spots_t = torch.as_tensor(_SPOTS, dtype=physics.dtype, device=dev)[:, :, None]      # slides are relative to the attachment frame
      bq = self._ball_q
      rows = [bq, bq + 1, bq + 2] + [self._q[p][k] for p in _PLAYERS for k in ('root_x', 'root_y')] + [self._q[p]['steer'] for p in _PLAYERS]
      self._place_rows = torch.as_tensor(rows, dtype=torch.long, device=dev)
      self._ctrl_rows_t = torch.as_tensor([r for rows_ in self._ctrl_rows for r in rows_], dtype=torch.long, device=dev)
    if self._randomize is not None:
      lo, hi = self._randomize
      size = self._uniform(physics, lo, hi)                    # one ratio per axis (keep_aspect_ratio=False)
      self._size_t.copy_(torch.where(mask[None, :], size, self._size_t))
      self._resize_pitch(physics, self._size_t, mask)
    self._place(physics, mask)
    self._prev_action.masked_fill_(mask[None, None, :], 0)
    physics.mark_as_dirty()

  def before_step(self, physics, action, random_state):
    """action: (B, 4, 3).  task.py:211-217: apply the players' actions, then throw the ball in if it left the court."""
    torch = physics.torch
    a = action.permute(1, 2, 0).to(physics.dtype)              # (4, 3, B)
    physics.field('ctrl').index_copy_(0, self._ctrl_rows_t, a.reshape(12, physics.B))
    self._prev_action.copy_(a)
    off = self.field.detected
    if off is not None:      # (None before the first initialize_episode)
      # _throw_in (task.py:128-135): ball back at a shrunk position, at rest
      q, v = physics.field('qpos'), physics.field('qvel')
      xy = self.ball_xpos(physics)[:2]
      shrink = self._uniform(physics, [0.7, 0.7], [0.9, 0.9])
      bq, bv = self._ball_q, self._ball_v
      q[bq:bq + 2] = torch.where(off[None, :], xy * shrink, q[bq:bq + 2])
      q[bq + 2].masked_fill_(off, _THROW_IN_BALL_Z - 0.35)
      v[bv:bv + 6].masked_fill_(off[None, :], 0)
      physics.mark_as_dirty()

  def _scoring_team(self, physics):
    """+1 where AWAY scored (ball in the home goal), -1 ... as two masks: (home_scored, away_scored)."""
    return self.away_goal.detected, self.home_goal.detected

  def get_reward(self, physics):
    torch = physics.torch
    home_scored, away_scored = self._scoring_team(physics)
    # arena.detected_goal: the home goal is looked at first (pitch.py:574-580)
    away_scored = away_scored
    home_scored = home_scored & ~away_scored
    r_home = home_scored.to(physics.dtype) - away_scored.to(physics.dtype)
    return torch.stack([r_home, r_home, -r_home, -r_home])                           # (4, B)

  def get_discount(self, physics):
    return (~(self.home_goal.detected | self.away_goal.detected)).to(physics.dtype)

  def should_terminate_episode(self, physics):
    return self.home_goal.detected | self.away_goal.detected

  # -- the control step as launches ------------------------------------------------------------------
  def device_step(self, env, action):
    """`composer.Environment.step` for this task on the device in six launches -- two random draws, `soccer_pre`, the
    physics launch with the substep probe, `soccer_post`, the observation gather -- instead of the ~150 small tensor
    operations of the hooks above (0.5 ms next to 0.9 ms of physics at B = 256).  Same random stream, same results
    (tests/test_gpu_composer.py::test_soccer_task_kernels_equal_the_tensor_task_layer).  Returns None where it does
    not apply (a randomised pitch, extra hooks, unfused substeps, a physics without device tensors): the caller then
    runs the hooks."""
    p = env.physics
    torch = p.torch
    if (not self._task_kernels or self._randomize is not None or not env.probed or self._gen is None or self.model.na or
        getattr(p, 'batch', None) is None or not hasattr(p.batch, 'set_step_probe') or p.device.type != 'cuda' or
        any(env._hooks._extra.values()) or os.environ.get('DMC_SOCCER_KERNELS', '1') == '0'):
      return None
    B, n_sub = p.B, env.n_sub_steps
    d = self.__dict__.get('_dev')
    if d is None or d['B'] != B or d['n_sub'] != n_sub:
      d = self._dev = self._device_plan(env)
    a = d['args']
    act = action if (action.dtype == p.dtype and action.is_contiguous() and action.device == p.device) else d['act'].copy_(action)
    a.action = act.data_ptr()
    torch.rand(d['u_place'].shape, generator=self._gen, device=p.device, dtype=p.dtype, out=d['u_place'])      # _place's draw
    torch.rand(d['u_shrink'].shape, generator=self._gen, device=p.device, dtype=p.dtype, out=d['u_shrink'])    # the throw-in's
    reward = torch.empty((4, B), dtype=p.dtype, device=p.device)
    discount = torch.empty(B, dtype=p.dtype, device=p.device)
    step_type = torch.empty(B, dtype=torch.int32, device=p.device)
    comp = torch.empty((B, 4, self._obs_C), dtype=p.dtype, device=p.device)
    a.reward, a.discount, a.step_type, a.comp = reward.data_ptr(), discount.data_ptr(), step_type.data_ptr(), comp.data_ptr()
    lib, st = d['lib'], p.stream()
    rc = lib.soccer_pre(st, ctypes.byref(a))
    if rc:
      raise RuntimeError('soccer_pre: hip error %d' % rc)
    p.mark_as_dirty()
    p.step(n_sub)
    env.launches += 1
    rc = lib.soccer_post(st, ctypes.byref(a))
    if rc:
      raise RuntimeError('soccer_post: hip error %d' % rc)
    G = p.gather(self._obs_table).view(B, 4, self._obs_P)
    return environment.TimeStep(step_type=step_type, reward=reward, discount=discount, observation=self._observation_views(G, comp))

  def _device_plan(self, env):
    p = env.physics
    torch = p.torch
    B, n_sub, m = p.B, env.n_sub_steps, self.model
    if getattr(self, '_obs_layout', None) is None:
      self._observation_plan(p)
    self.detectors.materialise(p)
    lib, S = task_kernels(32 if p.dtype == torch.float32 else 64)
    R = 4
    d = dict(B=B, n_sub=n_sub, lib=lib,
             u_place=torch.empty((R, 5, 3, B), dtype=p.dtype, device=p.device), u_shrink=torch.empty((2, B), dtype=p.dtype, device=p.device),
             act=torch.empty((B, 4, 3), dtype=p.dtype, device=p.device), qpos0=p.const(m.qpos0),
             trace=p.substep_probe(env._hooks.probe_geom, n_sub))
    a = S()
    a.B, a.nq, a.nv, a.nu, a.nsub, a.rounds = B, m.nq, m.nv, m.nu, n_sub, R
    a.ball_q, a.ball_v, a.ball_geom, a.ball_linvel = self._ball_q, self._ball_v, self._ball_geom, self._ball_linvel
    a.place_rows[:] = [int(r) for r in self._place_rows.tolist()]
    a.ctrl_rows[:] = [int(r) for r in self._ctrl_rows_t.tolist()]
    a.root[:] = self._root
    a.ncomp = self._obs_C
    a.time_limit = float(env._time_limit)
    a.spawn_ratio, a.ball_z, a.pi = self._spawn_ratio, _INIT_BALL_Z - 0.35, np.pi
    assert _THROW_IN_BALL_Z == _INIT_BALL_Z
    for name in ('qpos', 'qvel', 'ctrl', 'time', 'env_mode', 'warning', 'geom_xpos', 'xpos', 'xmat', 'cvel', 'sensordata'):
      setattr(a, name, p.field(name).data_ptr())
    a.warm = p.field('qacc_warmstart').data_ptr()
    a.qpos0, a.size = d['qpos0'].data_ptr(), self._size_t.data_ptr()
    a.state, a.lo, a.hi = self.detectors.state.data_ptr(), self.detectors.lo.data_ptr(), self.detectors.hi.data_ptr()
    a.prev_action, a.u_place, a.u_shrink = self._prev_action.data_ptr(), d['u_place'].data_ptr(), d['u_shrink'].data_ptr()
    a.trace, a.reset_next = d['trace'].data_ptr(), env._reset_next.data_ptr()
    d['args'] = a
    return d

  # -- observations --------------------------------------------------------------------------------------
  def _frame(self, physics, k):
    b = self._root[k]
    R = physics.field('xmat')[9*b:9*b + 9].reshape(3, 3, physics.B)
    return physics.field('xpos')[3*b:3*b + 3], R

  @staticmethod
  def _ego(vec, R):
    """world (n, B) vector in the frame R (3, 3, B): v . R restricted to the first n components."""
    n = vec.shape[0]
    return (vec[:, None, :] * R[:n, :n]).sum(dim=0)

  # What a player observes of the ball and of the others are the frame sensors CoreObservablesAdder adds to its model
  # (observables.py:96-198: framepos / framelinvel / frameangvel / frame?axis with reftype="body"): MuJoCo evaluates
  # objtype / reftype "body" in the bodies' INERTIAL frames (xipos, ximat -- the head's principal axes are a
  # permutation of its body axes) and velocities RELATIVE to the reference frame.  They are read, not recomputed
  # from xpos / xmat (round 2 did that, in the body frame: found by running the reference's task on this backend,
  # tests/test_reference_composer.py).
  _CORNER_NAMES = ('team_goal_back_right', 'team_goal_mid', 'team_goal_front_left', 'field_front_left',
                   'opponent_goal_back_left', 'opponent_goal_mid', 'opponent_goal_front_right', 'field_back_right')
  _CORNER_DIMS = (2, 3, 2, 2, 2, 3, 2, 2)
  _STATS = ('stats_vel_to_ball', 'stats_closest_vel_to_ball', 'stats_veloc_forward', 'stats_vel_ball_to_goal',
            'stats_home_avg_teammate_dist', 'stats_teammate_spread_out', 'stats_home_score', 'stats_away_score')

  def _observation_plan(self, physics):
    """Built once.  Everything an agent observes that is a row of an mjData field -- proprioception, its kinematic
    sensors, the ego-centric frame sensors of ball / teammate / opponents -- goes through ONE launch of the gather
    kernel (observation.GatherTable: a block of identical layout per player, so the (B, 4 P) matrix is the (B, 4, P)
    observation); the rest (goal / field corners in the player's frame, the stats_* scalars, prev_action) is computed
    for the four players at once.  Round 3 built the dictionary from ~600 per-player tensor operations: 60 % of the
    control step at B = 256."""
    from dm_control_amd.observation import GatherTable
    torch = physics.torch
    entries, layout = [], None
    for k, p in enumerate(_PLAYERS):
      mates = [j for j in range(4) if j != k and _TEAM[j] == _TEAM[k]]
      opps = [j for j in range(4) if _TEAM[j] != _TEAM[k]]
      mine = [('joints_pos', ('qpos', [p + '/kick'])), ('joints_vel', ('qvel', [p + '/kick'])),
              ('body_height', ('xpos', [p + '/head_body'], 'z')),
              ('end_effectors_pos', ('sensordata', [p + '/head_body_end_effector'])),
              ('world_zaxis', ('xmat', [p + '/head_body'], ['zx', 'zy', 'zz'])),
              ('sensors_gyro', ('sensordata', [p + '/sensor_torso_gyro'])),
              ('sensors_velocimeter', ('sensordata', [p + '/sensor_torso_vel'])),
              ('sensors_accelerometer', ('sensordata', [p + '/sensor_torso_accel'])),
              ('prev_action', None),
              ('ball_ego_position', ('sensordata', [p + '/ball_ego_pos'])),
              ('ball_ego_linear_velocity', ('sensordata', [p + '/ball_ego_linvel'])),
              ('ball_ego_angular_velocity', ('sensordata', [p + '/ball_ego_angvel']))]
      for prefix, others in (('teammate', mates), ('opponent', opps)):
        for n, j in enumerate(others):
          pre = '%s_%d' % (prefix, n)
          mine += [(pre + '_ego_end_effectors_pos', ('sensordata', ['%s/head_body_%s_end_effector' % (p, pre)])),
                   (pre + '_ego_linear_velocity', ('sensordata', ['%s/%s_ego_linear_velocity' % (p, pre)])),
                   (pre + '_ego_position', ('sensordata', ['%s/%s_ego_position' % (p, pre)])),
                   (pre + '_ego_orientation', ('sensordata', ['%s/%s_ego_orientation_%s' % (p, pre, d) for d in 'xyz'])),
                   # the other's end effectors in the OTHER's frame (observables.py:158-161: its own end_effectors_pos)
                   (pre + '_end_effectors_pos', ('sensordata', [_PLAYERS[j] + '/head_body_end_effector']))]
      mine += [(n, None) for n in self._CORNER_NAMES + self._STATS]
      table_entries = [e for _, e in mine if e is not None]
      widths = [GatherTable(self.model, [e]).size for e in table_entries]
      entries += table_entries
      if layout is None:      # name -> ('g', offset in the player's gathered block, width) | ('c', offset in the computed block, width)
        layout, go, it = [], 0, iter(widths)
        co = 0
        for name, e in mine:
          if e is not None:
            w = next(it); layout.append((name, 'g', go, w)); go += w
          else:
            w = 3 if name == 'prev_action' else (self._CORNER_DIMS[self._CORNER_NAMES.index(name)] if name in self._CORNER_NAMES else 1)
            if name == 'stats_veloc_forward':      # the first component of the velocimeter (observables.py:300-305)
              layout.append((name, 'v', 0, 1))
            else:
              layout.append((name, 'c', co, w)); co += w
        self._obs_P, self._obs_C = go, co
    self._obs_table = GatherTable(self.model, entries)
    assert self._obs_table.size == 4 * self._obs_P
    self._obs_layout = layout
    dev = physics.device
    self._root_t = torch.as_tensor(self._root, dtype=torch.long, device=dev)
    self._mate_t = torch.as_tensor([1, 0, 3, 2], dtype=torch.long, device=dev)
    self._team_t = torch.as_tensor(_TEAM, dtype=torch.bool, device=dev)[:, None]                # (4, 1): AWAY
    self._corner_idx = torch.as_tensor([list(range(8)) if t == 0 else [4, 5, 6, 7, 0, 1, 2, 3] for t in _TEAM], dtype=torch.long, device=dev)
    self._corner_keep = physics.const([[1.0, 1.0, 1.0 if d == 3 else 0.0] for d in self._CORNER_DIMS])[None, :, :, None]   # (1, 8, 3, 1)

  def get_observation(self, physics):
    torch = physics.torch
    B = physics.B
    if getattr(self, '_obs_layout', None) is None:
      self._observation_plan(physics)
    G = physics.gather(self._obs_table).view(B, 4, self._obs_P)
    nb = self.model.nbody
    pos = physics.field('xpos').view(nb, 3, B)[self._root_t]                     # (4, 3, B)
    R = physics.field('xmat').view(nb, 3, 3, B)[self._root_t]                    # (4, 3, 3, B): R[k, i, j]
    cv = physics.field('cvel').view(nb, 6, B)[self._root_t][:, 3:5]              # planar com-frame linear velocity of the roots
    ball = self.ball_xpos(physics)                                               # (3, B)
    sd = physics.field('sensordata')
    blin = sd[self._ball_linvel:self._ball_linvel + 3]
    home_lo, home_hi = self.home_goal.bounds(physics)
    away_lo, away_hi = self.away_goal.bounds(physics)
    home_mid, away_mid = (home_lo + home_hi) / 2, (away_lo + away_hi) / 2
    f_lo, f_hi = self.field.bounds(physics)
    z1 = torch.zeros((1, B), dtype=physics.dtype, device=physics.device)
    pad = lambda c: c if c.shape[0] == 3 else torch.cat([c[:2], z1])
    corners = torch.stack([pad(c) for c in (home_lo, home_mid, home_hi, f_hi, away_hi, away_mid, away_lo, f_lo)])   # (8, 3, B)
    # goal / field corners in the player's frame: (c - pos) . R over the corner's own dimensions (2-d corners: the planar block)
    D = (corners[self._corner_idx] - pos[:, None]) * self._corner_keep           # (4, 8, 3, B)
    E = torch.einsum('kmib,kijb->kmjb', D, R)
    # stats (observables.py:262-375)
    dir_ = ball[None] - pos
    planar = dir_[:, :2]
    unit = planar / (torch.linalg.norm(planar, dim=1, keepdim=True) + 1e-7)
    vel_to_ball = (unit * cv).sum(dim=1)                                         # (4, B)
    dist = torch.linalg.norm(dir_, dim=1)
    closest = dist <= dist[self._mate_t]
    goal_mid = torch.where(self._team_t[:, :, None], home_mid[None], away_mid[None])      # the goal a player attacks
    direction = goal_mid - ball[None]
    nrm = torch.linalg.norm(direction, dim=1, keepdim=True)
    ndir = torch.where(nrm > 0, direction / nrm.clamp_min(1e-30), direction)
    avg = torch.linalg.norm(pos - pos[self._mate_t], dim=1)
    # arena.detected_goal() (pitch.py:574-580): the home goal is looked at first -- the ball in it means AWAY scored
    away_goal_scored = self.home_goal.detected
    home_goal_scored = self.away_goal.detected & ~away_goal_scored
    mine = torch.where(self._team_t, away_goal_scored[None], home_goal_scored[None]).to(physics.dtype)
    theirs = torch.where(self._team_t, home_goal_scored[None], away_goal_scored[None]).to(physics.dtype)
    stats = torch.stack([vel_to_ball, torch.where(closest, vel_to_ball, torch.zeros_like(vel_to_ball)),
                         (ndir * blin[None]).sum(dim=1), avg, (avg > 5.).to(physics.dtype), mine, theirs], dim=1)   # (4, 7, B)
    pieces = [self._prev_action]                                                 # (4, 3, B)
    pieces += [E[:, m, :d] for m, d in enumerate(self._CORNER_DIMS)]
    comp = torch.cat(pieces + [stats], dim=1).permute(2, 0, 1)                   # (B, 4, C)
    return self._observation_views(G, comp)

  def _observation_views(self, G, comp):
    """The observation dictionary over the gathered (B, 4, P) and the computed (B, 4, C) blocks."""
    out = {}
    for name, src, off, w in self._obs_layout:
      if src == 'g':
        out[name] = G[:, :, off:off + w]
      elif src == 'c':
        out[name] = comp[:, :, off:off + w]
      else:
        out[name] = out['sensors_velocimeter'][:, :, :1]
    return out                                                                   # name -> (B, 4, n)


def make(batch_size, device_id=0, precision=32, time_limit=45.0, random_state=0, fuse_substeps=None, observation_options=None,
         delayed_observation_padding='zero', strip_singleton_obs_buffer_dim=True, **task_kwargs):
  """`soccer.load(team_size=2, time_limit=45., walker_type=WalkerType.BOXHEAD)` (soccer/__init__.py:92-148) for a
  batch, on the fixed pitch of the asset.  `observation_options` & co: composer/updater.py (buffered / delayed observables;
  the control step then runs through the tensor task layer, not the task kernels)."""
  task = Soccer2v2(**task_kwargs)
  physics = task.make_physics(batch_size, device_id=device_id, precision=precision)
  return environment.Environment(task, physics, time_limit=time_limit, random_state=random_state, fuse_substeps=fuse_substeps,
                                 observation_options=observation_options, delayed_observation_padding=delayed_observation_padding,
                                 strip_singleton_obs_buffer_dim=strip_singleton_obs_buffer_dim)
