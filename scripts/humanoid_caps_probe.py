"""Humanoid (BASELINE config 3) throughput vs contact / row caps, and the nefc / ncon actually reached."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
import torch
B = 4096
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/humanoid.xml')).read())
rs = np.random.RandomState(0)
for lanes, ncm, njm in ((64, 0, 0), (64, 12, 56), (64, 10, 44), (64, 8, 36), (32, 10, 44), (32, 8, 36)):
  try:
    b = BatchedPhysics(m, B, precision=32, lanes_per_env=lanes, nconmax=ncm, njmax=njm)
    q = np.tile(m.qpos0, (B, 1)); q[:, 7:] += rs.uniform(-0.2, 0.2, (B, m.nq - 7))
    b.set('qpos', q)
    b.set_output_mask(OUT['sensor'] | OUT['xpos'] | OUT['xmat'])
    T = 100
    ctrl = (torch.rand((T, m.nu, B), device='cuda', dtype=torch.float32) * 2 - 1)
    mx_efc = mx_con = 0
    for k in range(4):                      # 400 env-steps = 2000 physics steps: falls over and thrashes
      b.rollout(T, 5, ctrl.data_ptr(), None, None, None); b.sync()
      mx_efc = max(mx_efc, int(b.get('nefc').max())); mx_con = max(mx_con, int(b.get('ncon').max()))
    b.bind('ctrl', ctrl[0].data_ptr())
    ms = b.time_steps(5, 20)
    t0 = time.perf_counter(); b.rollout(T, 5, ctrl.data_ptr(), None, None, None); b.sync(); dt = time.perf_counter() - t0
    i = b.info()
    print(json.dumps(dict(lanes=lanes, nconmax=i['nconmax'], njmax=i['njmax'], envB=i['env_scratch_bytes'], epb=i['envs_per_block'],
                          lds=i['lds_bytes_per_block'], ms_env_step=ms, Menv_s=B / ms / 1e3, rollout_Menv_s=B * T / dt / 1e6,
                          max_nefc_seen=mx_efc, max_ncon_seen=mx_con, warnings=b.get('warning').sum(axis=0).tolist())), flush=True)
    b.close()
  except Exception as ex:
    print(json.dumps(dict(lanes=lanes, nconmax=ncm, njmax=njm, error=repr(ex))), flush=True)
