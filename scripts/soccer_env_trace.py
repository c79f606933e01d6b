"""A short soccer_2v2 environment loop for rocprofv3 --kernel-trace: which launches a control step is made of."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dm_control_amd import composer
B = int(os.environ.get('B', 256)); T = int(os.environ.get('T', 200))
env = composer.make('soccer_2v2', B, task_kernels=os.environ.get('KERNELS', '1') == '1')
env.reset()
gen = torch.Generator(device='cuda').manual_seed(0)
acts = torch.rand((16, B, 4, 3), device='cuda', generator=gen) * 2 - 1
for t in range(10):
  env.step(acts[t % 16])
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
  env.step(acts[t % 16])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('B %d: %.1f k env-steps/s, %.1f us per control step' % (B, B * T / dt / 1e3, dt / T * 1e6))
# physics only, same batch and outputs
p = env.physics
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
  p.step(env.n_sub_steps)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('physics launches only: %.1f us per control step' % (dt / T * 1e6))
