"""The ceiling of "a soccer match on several waves" (VERDICT r05 #2), measured instead of built.

One wave steps one match: 64 lanes split every loop over bodies / geoms / contacts / rows / matrix entries and meet at
wave-level fences (DMC_WSYNC: LDS operations of one wave retire in order, so a fence costs a few cycles).  W waves per
match would split the same loops over 64 W lanes -- and every fence between a producer loop and a consumer loop becomes
a workgroup barrier.  This script counts, on the host build of the kernel core (tests/emu: the same StepCore, one lane),
  * the fences one physics step of the BASELINE config-5 match executes, and
  * for every lane-split loop its trip count, i.e. how many of the 64 W lanes would have work,
over a seeded rollout; scripts/barrier_cost_probe.hip measures what a fence and a W-wave barrier cost on the device.
Both go to profiles/r06_multiwave_prototype.log (scripts/gpu_r06_multiwave.sh)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import emu_lib
from emu_lib import EmuPhysics
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.suite import common
from dm_control_amd.composer.tasks import soccer

m = mc.compile_xml(common.read_model('soccer_2v2_boxhead.xml'))
rs = np.random.RandomState(0)
nenv, T, nsub = 6, 40, 5
counts, iters, ncons = [], [], []
L = emu_lib.lib()
L.emu_wsync_count.restype = __import__('ctypes').c_longlong
for e in range(nenv):
  p = EmuPhysics(m, 32, nconmax=24)
  q = soccer.kickoff_qpos(m).copy()
  adr = soccer.addresses(m)
  q[[a for xy in adr['players'] for a in xy]] += rs.uniform(-8, 8, 8)
  q[adr['ball_q']:adr['ball_q'] + 2] += rs.uniform(-15, 15, 2)
  p.qpos[:] = q
  for t in range(T):
    p.ctrl[:] = rs.uniform(-1, 1, m.nu)
    for k in range(nsub):
      c0 = L.emu_wsync_count()
      p.step()
      counts.append(L.emu_wsync_count() - c0)
      iters.append(int(np.ravel(p.solver_iter)[0])); ncons.append(int(np.ravel(p.ncon)[0]))
counts = np.asarray(counts)
res = dict(model='soccer_2v2_boxhead', physics_steps=int(counts.size), fences_per_physics_step=dict(mean=float(counts.mean()), median=float(np.median(counts)), p90=float(np.percentile(counts, 90)), max=int(counts.max())),
           newton_iterations_per_step=float(np.mean(iters)), contacts_per_step=float(np.mean(ncons)))
print(json.dumps(res))
