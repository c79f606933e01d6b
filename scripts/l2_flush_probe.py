"""How long does a kernel that leaves N MB dirty in the L2s take back to back (launch + write + end-of-kernel write-back)?"""
import json, time, torch
out = {}
for mb in (0.25, 1, 2.9, 4, 8, 11.5, 16, 32, 64, 256):
  n = int(mb * (1 << 20) / 4)
  x = torch.empty(n, device='cuda', dtype=torch.float32)
  for _ in range(20): x.fill_(1.0)
  torch.cuda.synchronize()
  t = time.perf_counter()
  for i in range(300): x.fill_(float(i))
  torch.cuda.synchronize()
  us = (time.perf_counter() - t) / 300 * 1e6
  y = torch.empty_like(x)
  for _ in range(20): y.copy_(x)
  torch.cuda.synchronize()
  t = time.perf_counter()
  for i in range(300): y.copy_(x)
  torch.cuda.synchronize()
  usc = (time.perf_counter() - t) / 300 * 1e6
  out[str(mb)] = dict(fill_us=us, fill_GBps=mb * 1.048576e-3 / (us * 1e-6), copy_us=usc)
print(json.dumps(out))
