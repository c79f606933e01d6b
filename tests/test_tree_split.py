"""Factorisations block diagonal over the kinematic trees (StepDims::treemax, step_core.h chol_factor_trees /
chol_solve_trees / hess_assemble_split / h_split).

CPU tier: the tables, and the emulated kernel core, which under DMC_HOST_EMU checks on EVERY split solve that the full
Hessian has exact zeros wherever an entry joins two trees and that the in-tree assembly reproduces the full one bit for bit
(it aborts otherwise) -- on soccer 2v2 and on free boxes that lie apart (split) or collide (coupled).
`-m gpu`: a plugin kernel with the side-by-side routines against the generic kernel (full factorisations) on the boxes."""
import numpy as np
import pytest

import bench
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.suite import common
from emu_lib import EmuPhysics


def boxes(n=4, gap=0.6, friction=1.0):
  fr = 'friction="%g .005 .0001"' % friction
  body = ''.join('<body name="b%d" pos="%g 0 .1"><freejoint/><geom type="box" size=".1 .1 .1" mass="1" %s/></body>' % (k, gap*k, fr)
                 for k in range(n))
  return mc.compile_xml('<mujoco><option timestep="0.004"/><worldbody><geom name="floor" type="plane" size="5 5 .1" %s/>%s'
                        '</worldbody></mujoco>' % (fr, body))


def test_tables_of_the_soccer_model():
  m = bench.load_model('soccer_2v2_boxhead')
  t = EmuPhysics(m, 64).tree_tables()
  assert (t['treemax'], t['ntree'], t['ntreetri']) == (6, 5, 105)
  root = m.body_rootid[m.dof_bodyid]
  for i in range(m.nv):
    same = np.nonzero(root == root[i])[0]
    assert t['tree0'][i] == same.min() and t['tree1'][i] == same.max() + 1
  seen = set()
  for (i, j), ix in zip(t['tri'], t['trim']):
    assert i >= j and root[i] == root[j]
    seen.add((int(i), int(j)))
    anc, a = [], int(i)
    while a >= 0:
      anc.append(a); a = int(m.dof_parentid[a])
    assert (ix >= 0) == (int(j) in anc)      # M(i, j) is structurally nonzero exactly for ancestor pairs
  assert len(seen) == 105


def test_single_tree_and_small_models_do_not_split():
  for name in ('humanoid', 'cheetah', 'cmu_2019_position_floor'):
    assert EmuPhysics(bench.load_model(name), 64).tree_tables()['treemax'] == 0
  assert EmuPhysics(boxes(2), 64).tree_tables()['treemax'] == 0      # nv = 12: dense M, register Hessians
  assert EmuPhysics(boxes(3), 64).tree_tables()['treemax'] == 6


@pytest.mark.parametrize('prec', [64, 32])
def test_soccer_solves_split_and_the_split_assembly_equals_the_full_one(prec):
  cfg = bench.CONFIGS[5]
  m = bench.load_model(cfg['asset'])
  caps = {k: v for k, v in common.DEFAULT_CAPS.get(cfg['asset'], {}).items() if k in ('nconmax', 'njmax', 'njcon')}
  e = EmuPhysics(m, prec, **caps)
  e.qpos[:] = bench.initial_qpos(cfg, m, 2, 0)[1]
  rs = np.random.RandomState(0)
  n0 = EmuPhysics.split_solves()
  for _ in range(20):
    e.ctrl[:] = rs.uniform(-1, 1, m.nu)
    e.step(5)      # (aborts the process if a split solve's Hessian joins two trees or the in-tree assembly differs)
  assert EmuPhysics.split_solves() - n0 >= 90 and np.isfinite(e.qpos).all()


def test_boxes_apart_split_boxes_in_contact_do_not():
  m = boxes(3, gap=0.6)
  e = EmuPhysics(m, 64)
  e.set_islands(0)      # one joint solve per step (with islands on, each box is its own solve)
  n0 = EmuPhysics.split_solves()
  e.step(10)
  assert 8 <= EmuPhysics.split_solves() - n0 <= 10      # three boxes resting on the floor: every solve splits (a step without rows has no solve)
  # the first box slides into the second: while they touch, their contact's rows move two trees
  e.qvel[0] = 3.0
  n1 = EmuPhysics.split_solves()
  touched = 0
  for _ in range(150):
    e.step(1)
    g1, g2 = e.contact_geom1[:int(e.ncon[0])], e.contact_geom2[:int(e.ncon[0])]
    touched += int(np.any((g1 > 0) & (g2 > 0)))
  assert touched > 0
  assert EmuPhysics.split_solves() - n1 <= 150 - touched


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol', [(64, 1e-10), (32, 2e-5)])
def test_plugin_with_tree_routines_follows_the_generic_kernel(tmp_path, monkeypatch, precision, tol):
  from dm_control_amd.batch import BatchedPhysics
  monkeypatch.setenv('DMC_SPEC_CACHE', str(tmp_path))
  m = boxes(4, gap=0.45, friction=0.1)
  B = 32
  caps = dict(lanes_per_env=64, nconmax=40, njmax=200)
  g = BatchedPhysics(m, B, precision=precision, specialise='off', **caps)
  s = BatchedPhysics(m, B, precision=precision, specialise='build', **caps)
  assert s.specialised == 'attached' and g.info()['static_id'] == -1
  rs = np.random.RandomState(0)
  q = np.tile(m.qpos0, (B, 1))
  v = np.zeros((B, m.nv))
  v[:, 0] = rs.uniform(1, 3, B)      # the outer boxes slide at the inner ones: solves with and without a box-box contact
  v[:, 6 * 3] = -rs.uniform(1, 3, B)
  for b in (g, s):
    b.set('qpos', q); b.set('qvel', v)
  worst, pairs = 0.0, 0
  for _ in range(200):
    if precision == 32:      # fp32: step by step from the generic kernel's state
      for f in ('qpos', 'qvel', 'qacc_warmstart'):
        s.set(f, g.get(f))
    g.step(); s.step()
    worst = max(worst, float(np.abs(s.get('qpos') - g.get('qpos')).max()))
    pairs += int((g.get('ncon') > 16).sum())
  assert worst < tol, worst
  assert pairs > 0      # (four boxes on the floor have 16 contacts; more means two boxes touch)
  assert not s.get('warning').any()
  for b in (g, s):
    b.close()
