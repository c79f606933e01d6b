"""The large BASELINE models on the GPU: the two 62-dof CMU humanoids (suite humanoid_CMU: 10 substeps per env-step; BASELINE
config 4 physics -- position-controlled 2019 model on the Floor arena, elliptic cones + 5 noslip sweeps,
6 substeps) and BASELINE config 5 physics (soccer 2v2 BoxHead, 5 substeps): LDS geometry, throughput at B = 4096, contact statistics."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
from dm_control_amd.suite import common
import torch
B = int(os.environ.get('B', 4096))
rs = np.random.RandomState(0)
out = []
for name, nsub in (('humanoid_CMU', 10), ('cmu_2019_position_floor', 6), ('soccer_2v2_boxhead', 5)):
  m = mc.compile_xml(common.read_model(name + '.xml'))
  caps = dict(common.DEFAULT_CAPS[name])
  prec, ncon = caps.get('precision', 32), caps['nconmax']
  try:
    b = BatchedPhysics(m, B, precision=prec, nconmax=ncon)
    q = np.tile(m.qpos0, (B, 1))
    if name.startswith('soccer'):
      q[:, [0, 1, 6, 7, 12, 13, 18, 19]] += rs.uniform(-8, 8, (B, 8))     # players anywhere around their kick-off spots
      q[:, 24:26] += rs.uniform(-15, 15, (B, 2))
    else:
      q[:, 7:] += rs.uniform(-0.2, 0.2, (B, m.nq - 7))
    b.set('qpos', q)
    b.set_output_mask(OUT['sensor'] | OUT['xpos'] | OUT['xmat'] | OUT['subtree_com'])
    T = 50
    td = torch.float32 if prec == 32 else torch.float64
    ctrl = (torch.rand((T, m.nu, B), device='cuda', dtype=td) * 2 - 1)
    b.rollout(T, nsub, ctrl.data_ptr(), None, None, None); b.sync()   # fall to the floor
    b.bind('ctrl', ctrl[0].data_ptr())
    ms = b.time_steps(nsub, 5)
    t0 = time.perf_counter(); b.rollout(T, nsub, ctrl.data_ptr(), None, None, None); b.sync(); dt = time.perf_counter() - t0
    ncon_now = b.get('ncon')
    r = dict(model=name, nsub=nsub, prec=prec, nconmax=ncon, ms_per_env_step=ms, env_steps_per_s=B / (ms * 1e-3), physics_steps_per_s=B * nsub / (ms * 1e-3),
             rollout_env_steps_per_s=B * T / dt, info=b.info(), mean_ncon=float(ncon_now.mean()), max_ncon=int(ncon_now.max()),
             mean_iter=float(b.get('solver_iter').mean()), warnings=b.get('warning').sum(axis=0).tolist())
    b.close()
  except Exception as ex:  # pylint: disable=broad-except
    r = dict(model=name, nsub=nsub, prec=prec, nconmax=ncon, error=repr(ex))
  print(json.dumps(r), flush=True)
  out.append(r)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'cmu_probe.json'), 'w'), indent=1)
