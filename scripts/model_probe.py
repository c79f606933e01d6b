"""Throughput of suite models other than the headline config (humanoid: BASELINE
config 3; cartpole: RK4): env-steps/s at B=4096 with the task's n_sub_steps."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
import torch
B = int(os.environ.get('B', 4096))
out = []
_ALL = (('humanoid', 5, (64, 32)), ('cartpole', 1, (32, 16)), ('cheetah', 1, (32,)),
        ('walker', 10, (32,)), ('hopper', 4, (32,)))
_SEL = os.environ.get('MODELS')
for name, nsub, lanes_list in [x for x in _ALL if not _SEL or x[0] in _SEL.split(',')]:
  m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/%s.xml' % name)).read())
  rs = np.random.RandomState(0)
  for prec in (32, 64):
    for lanes in lanes_list:
      try:
        from dm_control_amd.suite import common
        b = BatchedPhysics(m, B, precision=prec, lanes_per_env=lanes, **common.DEFAULT_CAPS.get(name, {}))
        q = np.tile(m.qpos0, (B, 1))
        if name == 'humanoid':
          q[:, 7:] += rs.uniform(-0.2, 0.2, (B, m.nq - 7))
        elif name == 'cartpole':
          q += rs.uniform(-0.05, 0.05, q.shape)
        b.set('qpos', q)
        b.set_output_mask(OUT['sensor'] | OUT['xpos'] | OUT['xmat'] | OUT['subtree_com'])
        T = 100
        td = torch.float32 if prec == 32 else torch.float64
        ctrl = (torch.rand((T, m.nu, B), device='cuda', dtype=td) * 2 - 1)
        b.rollout(T, nsub, ctrl.data_ptr(), None, None, None); b.sync()   # settle / fall
        b.bind('ctrl', ctrl[0].data_ptr())
        ms = b.time_steps(nsub, 20)
        t0 = time.perf_counter(); b.rollout(T, nsub, ctrl.data_ptr(), None, None, None); b.sync(); dt = time.perf_counter() - t0
        r = dict(model=name, prec=prec, lanes=lanes, nsub=nsub, ms_per_env_step=ms, env_steps_per_s=B / (ms * 1e-3),
                 physics_steps_per_s=B * nsub / (ms * 1e-3), rollout_env_steps_per_s=B * T / dt, info=b.info(),
                 mean_ncon=float(b.get('ncon').mean()), mean_iter=float(b.get('solver_iter').mean()),
                 warnings=b.get('warning').sum(axis=0).tolist())
        b.close()
      except Exception as ex:  # pylint: disable=broad-except
        r = dict(model=name, prec=prec, lanes=lanes, error=repr(ex))
      print(json.dumps(r), flush=True)
      out.append(r)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', os.environ.get('OUTNAME', 'model_probe.json')), 'w'), indent=1)
