"""Which source lines of the composer task layer launch the control step's kernels (soccer_2v2, B = 256, eager): device
kernel launches grouped by the innermost frame inside dm_control_amd/ (torch.profiler with stacks)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from dm_control_amd import composer
env = composer.make(os.environ.get('ENV', 'soccer_2v2'), int(os.environ.get('B', 256)))
env.reset()
gen = torch.Generator(device='cuda').manual_seed(0)
acts = torch.rand((16, int(os.environ.get('B', 256)), 4, 3), device='cuda', generator=gen) * 2 - 1
for t in range(5): env.step(acts[t])
torch.cuda.synchronize()
N = 10
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
sites, ops = collections.Counter(), collections.Counter()
class Count(TorchDispatchMode):
  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    name = str(func)
    fr = None
    for f in reversed(traceback.extract_stack(limit=14)):
      if 'dm_control_amd' in f.filename and 'scripts' not in f.filename:
        fr = '%s:%d %s' % (f.filename.split('dm_control_amd/')[-1], f.lineno, f.name)
        break
    if fr is not None and 'view' not in name and 'reshape' not in name and 'slice' not in name and 'select' not in name and 'alias' not in name \
        and 'permute' not in name and 'expand' not in name and 't.default' not in name and 'unsqueeze' not in name and 'squeeze' not in name and 'detach' not in name and 'transpose' not in name and 'as_strided' not in name and 'unbind' not in name and 'split' not in name:
      sites[fr] += 1; ops[name] += 1
    return func(*args, **(kwargs or {}))
with Count():
  for t in range(N): env.step(acts[t % 16])
torch.cuda.synchronize()
print('device-launching aten ops per step (views excluded): %.1f' % (sum(sites.values()) / N))
byfn = collections.Counter()
for k, v in sites.items():
  byfn[k.split(' ')[0].split(':')[0] + ' ' + k.split(' ')[-1]] += v
for k, v in byfn.most_common(25): print('  %-60s %.1f per step' % (k, v / N))
print({k: round(v / N, 1) for k, v in ops.most_common(18)})
