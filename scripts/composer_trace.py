"""Kernel trace of the soccer environment's control step (B = 256): which kernels a step launches and how long they run.
Run under rocprofv3 --kernel-trace --stats (scripts/gpu_r05_composer_trace.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dm_control_amd import composer
env = composer.make(os.environ.get('ENV', 'soccer_2v2'), int(os.environ.get('B', 256)))
B = env.physics.batch.batch_size if hasattr(env.physics, 'batch') else int(os.environ.get('B', 256))
env.reset()
gen = torch.Generator(device='cuda').manual_seed(0)
acts = torch.rand((16, int(os.environ.get('B', 256)), 4, 3), device='cuda', generator=gen) * 2 - 1
mode = os.environ.get('MODE', 'eager')
for t in range(5): env.step(acts[t])
if mode == 'graph': env.capture(acts[0])
step = env.step_graph if mode == 'graph' else env.step
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
N = 200
for t in range(N): step(acts[t % 16])
torch.cuda.synchronize()
print('ms per env-step', 1e3 * (time.perf_counter() - t0) / N)
