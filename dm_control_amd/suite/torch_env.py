"""Device-resident batched environments.

`control.Environment` over the host mirror costs a PCIe round trip per field per
step.  `TorchBatchedEnv` keeps everything on the GPU: the SoA arrays of the HIP
batch are rebound to torch tensors (zero copy, `dmc_batch_bind`), actions are
written by the caller's policy directly into the `ctrl` tensor, observations /
rewards are torch expressions over the `qpos` / `qvel` / `sensordata` tensors,
and environments whose episode ended are re-initialised on device from a pool of
pre-settled start states.  torch is plumbing here (memory + elementwise ops); the
physics is the fused HIP kernel.

Only the cheetah `run` task is provided in this form (BASELINE config 2).
"""
import numpy as np

from dm_control_amd import mjcf_compiler
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common

_RUN_SPEED = 10.0


class TorchBatchedEnv:
  """Cheetah run, B environments, all tensors (rows, B) on `device`."""

  def __init__(self, batch_size, device_id=0, precision=32, time_limit=10.0, seed=0, n_sub_steps=1):
    import torch
    self.torch = torch
    self.device = torch.device('cuda', device_id)
    self.model = mjcf_compiler.compile_xml(common.read_model('cheetah.xml'))
    m = self.model
    self.B = int(batch_size)
    self.n_sub_steps = n_sub_steps
    self.dtype = torch.float32 if precision == 32 else torch.float64
    self.physics = BatchedPhysics(m, self.B, device_id=device_id, precision=precision)
    self.physics.set_output_mask(OUT['sensor'])
    mk = lambda rows: torch.zeros((rows, self.B), dtype=self.dtype, device=self.device)
    self.qpos, self.qvel, self.ctrl = mk(m.nq), mk(m.nv), mk(m.nu)
    self.warm, self.sensordata = mk(m.nv), mk(m.nsensordata)
    self.time = torch.zeros((1, self.B), dtype=torch.float64, device=self.device)
    for name, t in (('qpos', self.qpos), ('qvel', self.qvel), ('ctrl', self.ctrl),
                    ('qacc_warmstart', self.warm), ('sensordata', self.sensordata), ('time', self.time)):
      self.physics.bind(name, t.data_ptr())
    self.step_limit = int(round(time_limit / (m.opt.timestep * n_sub_steps)))
    self.steps = torch.zeros(self.B, dtype=torch.int64, device=self.device)
    self._rs = np.random.RandomState(seed)
    self._make_start_pool()
    self.reset()

  def _make_start_pool(self, pool=None):
    """Cheetah.initialize_episode for a pool of start states (suite/cheetah.py:63-76):
    limited joints ~ U(range), 200 settle steps with zero control, on device."""
    torch, m = self.torch, self.model
    pool = pool or self.B
    assert pool == self.B
    lim = m.jnt_limited == 1
    lo, hi = m.jnt_range[lim].T
    q = np.tile(m.qpos0, (self.B, 1))
    q[:, lim] = self._rs.uniform(lo, hi, (self.B, lo.size))
    self.qpos.copy_(torch.from_numpy(q.T.copy()).to(self.dtype))
    self.qvel.zero_(); self.ctrl.zero_(); self.warm.zero_()
    self.physics.step(200, stream=torch.cuda.current_stream().cuda_stream)
    self.pool_qpos, self.pool_qvel, self.pool_warm = self.qpos.clone(), self.qvel.clone(), self.warm.clone()

  def reset(self, mask=None):
    """Re-initialises the selected environments (all if mask is None) from the pool."""
    torch = self.torch
    if mask is None:
      mask = torch.ones(self.B, dtype=torch.bool, device=self.device)
    perm = torch.randperm(self.B, device=self.device)   # start state drawn from the pool
    m2 = mask[None, :]
    self.qpos.copy_(torch.where(m2, self.pool_qpos[:, perm], self.qpos))
    self.qvel.copy_(torch.where(m2, self.pool_qvel[:, perm], self.qvel))
    self.warm.copy_(torch.where(m2, self.pool_warm[:, perm], self.warm))
    self.time.copy_(torch.where(m2, torch.zeros_like(self.time), self.time))
    self.steps.copy_(torch.where(mask, torch.zeros_like(self.steps), self.steps))
    return self.observation()

  def observation(self):
    """(B, 17): qpos[1:] and qvel (Cheetah.get_observation)."""
    return self.torch.cat([self.qpos[1:], self.qvel], dim=0).T

  def reward(self):
    """rewards.tolerance(speed, bounds=(10, inf), margin=10, value_at_margin=0, 'linear')."""
    speed = self.sensordata[0]
    return self.torch.clamp(speed / _RUN_SPEED, 0.0, 1.0)

  def step(self, action):
    """action: (B, nu) tensor on device.  Returns (obs, reward, done) tensors; finished
    environments are auto-reset (their returned obs is the fresh start state)."""
    torch = self.torch
    self.ctrl.copy_(action.T.to(self.dtype))
    self.physics.step(self.n_sub_steps, stream=torch.cuda.current_stream().cuda_stream)
    self.steps += 1
    reward = self.reward().clone()
    done = self.steps >= self.step_limit
    obs = self.observation()
    if bool(done.any()):
      obs = self.reset(done)
    return obs, reward, done

  def close(self):
    self.physics.close()
