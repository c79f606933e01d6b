"""Row a1 (`Environment.step`: action -> physics -> observation, reward, restart) for every suite task, device-resident.

  * configs 2 / 3 through the hand-written task layers of suite/torch_env.py (per-environment auto-reset, randomisation
    drawn on the device): cheetah-run and humanoid-stand, B = 4096, 1000 env-steps;
  * all 45 tasks through suite/device_env.py (the host ports' task code on device tensors, control step as a captured HIP
    graph, whole-batch restart at the time limit): B = 4096, env-steps/s next to the physics-only rate of the same launch
    (ctrl write + dmc_batch_step, no task layer) measured in the same process.

Writes gpurun_out/r05_config_runs.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dm_control_amd import suite
from dm_control_amd.suite import device_env, torch_env

B = int(os.environ.get('B', 4096))
out = {'torch_env': [], 'device_env': []}
for domain, task, T, graph in (('cheetah', 'run', 1000, False), ('cheetah', 'run', 1000, True), ('humanoid', 'stand', 1000, False),
                               ('humanoid', 'stand', 1000, True)):
  env = torch_env.make(domain, task, B, precision=32, seed=0, capture=graph)
  nu = env.model.nu
  g = torch.Generator(device='cuda').manual_seed(0)
  acts = torch.rand((100, B, nu), device='cuda', generator=g) * 2 - 1
  for t in range(20):
    env.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(T):
    obs, rew, done = env.step(acts[t % 100])
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  # physics only, same batch object: the launch without the task layer
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  for t in range(200):
    env.ctrl.copy_(acts[t % 100].T)
    env.physics.step(env.n_sub_steps, stream=torch.cuda.current_stream().cuda_stream)
  torch.cuda.synchronize()
  dp = (time.perf_counter() - t1) / 200
  r = dict(task='%s %s' % (domain, task), mode='HIP graph' if graph else 'eager', B=B, env_steps=T, n_sub_steps=env.n_sub_steps, env_steps_per_s=B * T / dt,
           physics_only_env_steps_per_s=B / dp, ratio=(B * T / dt) / (B / dp), obs_finite=bool(torch.isfinite(obs).all()),
           warnings=env.physics.get('warning').sum(axis=0).tolist())
  print(json.dumps(r), flush=True)
  out['torch_env'].append(r)
  env.close()

only = os.environ.get('TASKS')
for domain, task in sorted(suite.ALL_TASKS):
  if only and ('%s-%s' % (domain, task)) not in only.split(','):
    continue
  try:
    env = device_env.make(domain, task, B, precision=32, seed=0, capture=True)
    nu = env.model.nu
    g = torch.Generator(device='cuda').manual_seed(0)
    acts = torch.rand((50, B, nu), device='cuda', generator=g) * 2 - 1
    for t in range(5):
      env.step(acts[t])
    torch.cuda.synchronize()
    T = 200
    t0 = time.perf_counter()
    for t in range(T):
      obs, rew, done = env.step(acts[t % 50])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / T
    t1 = time.perf_counter()
    for t in range(T):
      env.ctrl.copy_(acts[t % 50].T)
      env.host_physics.batch.step(env.n_sub_steps, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dp = (time.perf_counter() - t1) / T
    w = env.warnings().sum(axis=0).tolist()
    r = dict(task='%s %s' % (domain, task), B=B, n_sub_steps=env.n_sub_steps, nobs=int(obs.shape[1]), env_steps_per_s=B / dt,
             physics_only_env_steps_per_s=B / dp, ratio=dp / dt, obs_finite=bool(torch.isfinite(obs).all()),
             mean_reward=float(rew.mean()), warnings=w)
    env.close()
  except Exception as e:      # pylint: disable=broad-except
    r = dict(task='%s %s' % (domain, task), error='%s: %s' % (type(e).__name__, str(e)[:300]))
  print(json.dumps(r), flush=True)
  out['device_env'].append(r)
ok = [r for r in out['device_env'] if 'error' not in r]
out['summary'] = dict(tasks=len(out['device_env']), ok=len(ok), min_ratio=min((r['ratio'] for r in ok), default=None),
                      median_ratio=float(np.median([r['ratio'] for r in ok])) if ok else None)
print(json.dumps(out['summary']))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r05_config_runs.json'), 'w'), indent=1)
