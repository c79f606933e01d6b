import json, os, sys, time
import numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT)
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics, OUT
m = mc.compile_xml(open(os.path.join(ROOT, 'dm_control_amd/suite/assets/cheetah.xml')).read())
B=4096
lim = m.jnt_limited == 1
lo, hi = m.jnt_range[lim].T
q0 = np.tile(m.qpos0, (B, 1))
for e in range(B): q0[e, lim] = np.random.RandomState(e).uniform(lo, hi)
rs = np.random.RandomState(5)
b = BatchedPhysics(m, B, precision=32)
b.set('qpos', q0); b.set_output_mask(OUT['sensor']); b.step(200); b.sync()
b.set_control(rs.uniform(-1, 1, (B, m.nu)))
res={}
res['time_steps_100']=[b.time_steps(1,100) for _ in range(4)]
res['time_steps_8']=[b.time_steps(1,8) for _ in range(4)]
def pyloop(n):
  b.sync(); t=time.perf_counter()
  for _ in range(n): b.step()
  b.sync(); return (time.perf_counter()-t)/n*1e3
res['pyloop_100']=[pyloop(100) for _ in range(3)]
res['pyloop_8']=[pyloop(8) for _ in range(3)]
b.wave_trace(True)
def traced(n):
  b.sync()
  for _ in range(n): b.step()
  b.sync()
  tr=b.wave_trace().astype(np.int64)
  ent=tr[:,0].min(axis=1); en=tr[:,2].max(axis=1); st=tr[:,1]
  order=np.argsort(ent)
  ent=ent[order]; en=en[order]
  return dict(period=np.diff(ent).tolist(), span=(en-ent).tolist(), dur_med=[float(np.median(tr[k,2]-tr[k,1])) for k in order])
res['traced_8']=traced(8)
res['traced_100']=traced(100)
res['traced_1000']=traced(1000)
print(json.dumps(res))
b.sync()
for _ in range(8): b.step()
b.sync()
tr=b.wave_trace().astype(np.int64)
it=b.get('solver_iter')[:,0]; nc=b.get('ncon')[:,0]
for k in range(8):
  ent0=tr[k,0].min(); dur=tr[k,2]-tr[k,1]; top=np.argsort(-dur)[:6]
  print('launch',k,'span',int(tr[k,2].max()-ent0),[(int(i),int(dur[i]),int(tr[k,1,i]-ent0),int(tr[k,3,i]),int(tr[k,3,i])%8, int(it[2*i]),int(it[2*i+1]),int(nc[2*i]),int(nc[2*i+1])) for i in top])
dur=(tr[:,2]-tr[:,1])
print('per-item mean dur corr between launches', np.corrcoef(dur[0],dur[1])[0,1], np.corrcoef(dur[2],dur[5])[0,1])
print('dur hist', np.histogram(dur[3], bins=12))
# by solver iterations
d3=dur[3]; key=np.maximum(it[0::2],it[1::2])
for v in np.unique(key): print('iter',v,'n',int((key==v).sum()),'mean dur',float(d3[key==v].mean()),'max',int(d3[key==v].max()))
