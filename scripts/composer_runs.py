"""BASELINE configs 4 and 5 as device-resident environments (dm_control_amd.composer): env-steps/s over 1000 control
steps with random actions, auto-reset, observations + rewards evaluated every step, no host sync in the loop."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dm_control_amd import composer

out = []
T = int(os.environ.get('T', 1000))
ONLY = os.environ.get('ONLY')
for name, B, kw in (('cmu_go_to_target', 4096, {}), ('soccer_2v2', 256, {}), ('soccer_2v2', 256, dict(task_kernels=False)),
                    ('soccer_2v2', 256, dict(fuse_substeps=True)), ('soccer_2v2', 4096, {}), ('soccer_2v2', 4096, dict(task_kernels=False)),
                    ('soccer_2v2', 4096, dict(fuse_substeps=True))):
  if ONLY and ONLY not in name:
    continue
  env = composer.make(name, B, **kw)
  m = env.task.model
  shape = (B, 4, 3) if name.startswith('soccer') else (B, m.nu)
  gen = torch.Generator(device='cuda').manual_seed(0)
  env.reset()
  acts = torch.rand((16,) + shape, device='cuda', generator=gen) * 2 - 1
  for t in range(5):
    env.step(acts[t % 16])
  torch.cuda.synchronize()
  for mode in (('eager', 'graph') if os.environ.get('GRAPH') else ('eager',)):
   nlast = torch.zeros((), device='cuda', dtype=torch.int64)
   rsum = torch.zeros((), device='cuda', dtype=torch.float64)
   if mode == 'graph':
     env.capture(acts[0])
   stepfn = env.step_graph if mode == 'graph' else env.step
   torch.cuda.synchronize()
   t0 = time.perf_counter()
   for t in range(T):
     ts = stepfn(acts[t % 16])
     nlast += (ts.step_type == composer.LAST).sum()
     rsum += ts.reward.sum()
   torch.cuda.synchronize()
   dt = time.perf_counter() - t0
   r = dict(env=name, B=B, kwargs=kw, mode=mode, steps=T, n_sub_steps=env.n_sub_steps, fused=env.fused, seconds=dt,
           env_steps_per_s=B * T / dt, physics_steps_per_s=B * T * env.n_sub_steps / dt, episodes_ended=int(nlast),
           reward_sum=float(rsum), warnings=env.physics.field('warning').sum(dim=1).tolist(), info=env.physics.batch.info())
   print(json.dumps(r), flush=True)
   out.append(r)
  env.close()
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'composer_runs.json'), 'w'), indent=1)
