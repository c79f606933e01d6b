"""BASELINE configs 2 and 3 end to end on device (suite/torch_env.py): B=4096 environments, 1000 env-steps
of random actions including observations, rewards and auto-resets; prints env-steps/s and health checks."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm_control_amd.suite import torch_env

out = []
for domain, task, T in (('cheetah', 'run', 1000), ('humanoid', 'stand', 1000)):
  B = 4096
  env = torch_env.make(domain, task, B, precision=32, seed=0)
  nu = env.model.nu
  g = torch.Generator(device='cuda').manual_seed(0)
  acts = torch.rand((100, B, nu), device='cuda', generator=g) * 2 - 1
  for t in range(20):
    env.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  rsum = torch.zeros(B, device='cuda')
  ndone = 0
  for t in range(T):
    obs, rew, done = env.step(acts[t % 100])
    rsum += rew.to(rsum.dtype)
    ndone += int(done.sum()) if t % 100 == 99 else 0
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  w = env.physics.get('warning').sum(axis=0).tolist()
  r = dict(config='%s %s' % (domain, task), B=B, env_steps=T, n_sub_steps=env.n_sub_steps, seconds=dt,
           env_steps_per_s=B * T / dt, physics_steps_per_s=B * T * env.n_sub_steps / dt,
           obs_shape=list(obs.shape), obs_finite=bool(torch.isfinite(obs).all()), mean_reward=float(rsum.mean() / T),
           warnings=w)
  print(json.dumps(r), flush=True)
  out.append(r)
  env.close()
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'config_runs.json'), 'w'), indent=1)
