"""Row a1 for every suite task through suite/fused_env.py (the task layer as one generated kernel, per-environment
restarts on the device): env-steps/s next to the physics-only rate of the same batch (action write + dmc_batch_step, no
task layer), B = 4096, fp32.  Writes gpurun_out/r06_fused_env_runs.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dm_control_amd import suite
from dm_control_amd.suite import fused_env

B = int(os.environ.get('B', 4096))
T = int(os.environ.get('T', 300))
only = os.environ.get('TASKS')
out = []
for domain, task in sorted(suite.ALL_TASKS):
  if only and ('%s-%s' % (domain, task)) not in only.split(','):
    continue
  try:
    t_make = time.perf_counter()
    env = fused_env.make(domain, task, B, precision=32, seed=0, capture=True, copy_outputs=False)
    t_make = time.perf_counter() - t_make
    nu = env.model.nu
    g = torch.Generator(device='cuda').manual_seed(0)
    acts = torch.rand((50, B, nu), device='cuda', generator=g) * 2 - 1
    # both loops start from freshly restarted episodes (the cost of a physics step depends on where in the episode the
    # batch is: a finger that has flung its spinner away steps 2 x faster than one pushing it) and walk the same actions
    ctrl = env._tensors['ctrl']
    stream = torch.cuda.current_stream().cuda_stream
    env.restart(); env.step(acts[0])
    keep = [env.pending, env.steps, env.episode, env._tensors['env_mode']] + [env._tensors[f] for f in env._state_names] + list(env._attr_live.values())
    saved = [t.clone() for t in keep]
    for t in range(1, 6):
      ctrl.copy_(acts[t].T); env.host_physics.batch.step(env.n_sub_steps, stream=stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for t in range(T):
      ctrl.copy_(acts[t % 50].T)
      env.host_physics.batch.step(env.n_sub_steps, stream=stream)
    torch.cuda.synchronize()
    dp = (time.perf_counter() - t1) / T
    for t_, v_ in zip(keep, saved):      # the env loop starts from the very state the physics-only loop started from
      t_.copy_(v_)
    for t in range(1, 6):
      env.step(acts[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(T):
      obs, rew, done = env.step(acts[t % 50])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / T
    r = dict(task='%s %s' % (domain, task), B=B, n_sub_steps=env.n_sub_steps, nobs=int(obs.shape[1]), nodes=env.program.n_nodes,
             env_steps_per_s=B / dt, physics_only_env_steps_per_s=B / dp, ratio=dp / dt, us_per_step=1e6 * dt, us_physics=1e6 * dp,
             obs_finite=bool(torch.isfinite(obs).all()), mean_reward=float(rew.mean()), episodes=int(env.episode.sum().item()),
             warnings=env.warnings().sum(axis=0).tolist(), make_seconds=t_make, inline=env.inline)
    env.close()
  except Exception as e:      # pylint: disable=broad-except
    r = dict(task='%s %s' % (domain, task), error='%s: %s' % (type(e).__name__, str(e)[:300]))
  print(json.dumps(r), flush=True)
  out.append(r)
ok = [r for r in out if 'error' not in r]
summary = dict(tasks=len(out), ok=len(ok), min_ratio=min((r['ratio'] for r in ok), default=None),
               median_ratio=float(np.median([r['ratio'] for r in ok])) if ok else None,
               below_0_9=[r['task'] for r in ok if r['ratio'] < 0.9])
print(json.dumps(summary))
json.dump(dict(runs=out, summary=summary), open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r06_fused_env_runs.json'), 'w'), indent=1)
