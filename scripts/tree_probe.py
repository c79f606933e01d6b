"""How long is the step of ONE kinematic tree of the soccer model?  Sub-models cut out of the config-5 asset (one BoxHead
player / the ball, each with the whole static pitch) against the full 30-dof model, same kernel family (DMC_NO_STATIC=1
puts the full model on the generic kernel too).  ms per launch of 5 substeps at B = 256."""
import json, os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xml.etree.ElementTree as ET
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics
from dm_control_amd.suite import common

xml = common.read_model('soccer_2v2_boxhead.xml')


def cut(keep):
  """keeps the movable top-level bodies named in `keep`, drops the other ones with everything that refers to them"""
  root = ET.fromstring(xml)
  wb = root.find('worldbody')
  movable = ['soccer_ball/', 'home0/', 'home1/', 'away0/', 'away1/']
  drop = [n for n in movable if n not in keep]
  for b in list(wb):
    if b.tag == 'body' and b.get('name') in drop:
      wb.remove(b)
  def gone(v):
    return v is not None and any(v.startswith(d) for d in drop)
  for sec in ('actuator', 'sensor', 'contact', 'equality', 'tendon'):
    for s in root.findall(sec):
      for e in list(s):
        if any(gone(v) for v in e.attrib.values()):
          s.remove(e)
  return ET.tostring(root, encoding='unicode')


def run(name, x, B=256, nsub=5, caps=None):
  m = mc.compile_xml(x)
  b = BatchedPhysics(m, B, precision=32, **(caps or {}))
  rs = np.random.RandomState(0)
  q = np.tile(m.qpos0, (B, 1))
  # spread over the pitch like bench.initial_qpos does for config 5: x, y of every root
  for j in range(m.njnt):
    if m.jnt_type[j] == 0:
      a = m.jnt_qposadr[j]; q[:, a] = rs.uniform(-15, 15, B); q[:, a + 1] = rs.uniform(-10, 10, B); q[:, a + 2] = 0.5
  names = m.names['joint']
  for j, n in enumerate(names):
    if n and n.endswith('root_x'): q[:, m.jnt_qposadr[j]] = rs.uniform(-15, 15, B)
    if n and n.endswith('root_y'): q[:, m.jnt_qposadr[j]] = rs.uniform(-10, 10, B)
  b.set('qpos', q)
  b.forward(); b.sync()
  for t in range(60):
    b.set_control(rs.uniform(-1, 1, (B, m.nu))); b.step(nsub)
  b.sync()
  ms = min(b.time_steps(nsub, 20) for _ in range(3))
  info = b.info()
  out = dict(name=name, nv=int(m.nv), nbody=int(m.nbody), ngeom=int(m.ngeom), B=B, ms_per_launch=ms, lanes=info['lanes_per_env'],
             static_id=info['static_id'], envs_per_block=info['envs_per_block'], mean_ncon=float(b.get('ncon').mean()),
             mean_nefc=float(b.get('nefc').mean()), mean_iter=float(b.get('solver_iter').mean()), warnings=[int(w) for w in b.get('warning').sum(axis=0)])
  print(json.dumps(out), flush=True)
  b.close()
  return out


res = []
caps = dict(common.DEFAULT_CAPS.get('soccer_2v2_boxhead', {}))
res.append(run('full', xml, caps=caps))
for keep in (['home0/'], ['soccer_ball/'], ['home0/', 'soccer_ball/'], ['home0/', 'home1/']):
  for lanes in (64, 32, 16):
    try:
      res.append(run('+'.join(keep) + ' lanes%d' % lanes, cut(keep), caps=dict(lanes_per_env=lanes)))
    except Exception as e:
      print('failed', keep, lanes, repr(e)[:200], flush=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'tree_probe.json'), 'w'), indent=1)
