"""Writes suite/assets/cmu_2019_position_floor.xml: the physics of BASELINE config 4.

The reference builds that model at run time with PyMJCF (locomotion/examples/basic_cmu_2019.py:97-118):
`composer/arena.xml` (elliptic cones, 5 noslip sweeps, radians) + `arenas/floors.py` Floor (8 x 8 plane)
+ `walkers/cmu_humanoid.py` CMUHumanoidPositionControlled (humanoid_CMU_V2019.xml attached through a
free joint, motors replaced by scaled position actuators, `scaled_actuators.py:71-81`), physics
timestep 0.005 s.  PyMJCF is not importable here, so this script restates the composition from the
three sources; rendering-only elements are dropped, names lose their `walker/` prefix.  The
free-joint frame is placed at the walker's upright pose (cmu_humanoid.py:174-176), i.e. qpos0 is the
state UprightInitializer produces.  Only runnable where the reference tree exists."""
import ast
import os
import re
import sys
import xml.etree.ElementTree as ET

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import restate_model

REF = '/root/reference/dm_control/locomotion'


def position_actuators():
  src = open(os.path.join(REF, 'walkers/cmu_humanoid.py')).read()
  block = src[src.index('_POSITION_ACTUATORS = ['):]
  block = block[:block.index(']\n') + 1]
  out = []
  for name, lo, hi, kp in re.findall(r"PositionActuatorParams\('(\w+)',\s*\[\s*(-?[\d.]+),\s*(-?[\d.]+)\s*\],\s*([\d.]+)\s*\)", block):
    out.append((name, float(lo), float(hi), float(kp)))
  assert len(out) == 56, len(out)
  return out


def main(dst):
  root = ET.fromstring(open(os.path.join(REF, 'walkers/assets/humanoid_CMU_V2019.xml')).read())
  restate_model.strip(root)
  for e in root.findall('size'):
    root.remove(e)
  # arena: compiler / option of composer/arena.xml, timestep of the example
  comp = ET.Element('compiler', dict(coordinate='local', angle='radian', eulerseq='xyz', boundmass='1e-5', boundinertia='1e-11'))
  opt = ET.Element('option', dict(cone='elliptic', noslip_iterations='5', noslip_tolerance='0', timestep='0.005'))
  root.insert(0, comp)
  root.insert(1, opt)
  wb = root.find('worldbody')
  walker_root = wb.find('body')
  wb.remove(walker_root)
  for e in list(wb):      # lights / cameras are gone; nothing else lives in the walker's worldbody
    wb.remove(e)
  # PyMJCF scopes the walker's <default> to the walker; the floor keeps MuJoCo's built-in geom defaults
  # (mjcf/schema.xml:311-347), written out here because this file has a single default tree
  wb.append(ET.Element('geom', dict(name='groundplane', type='plane', size='8 8 0.25', condim='3',
                                    friction='1 0.005 0.0001', solref='0.02 1', solimp='0.9 0.95 0.001 0.5 2')))
  frame = ET.SubElement(wb, 'body', dict(name='walker', pos='0 0 0.94', quat='0.859 1 1 0.859'))
  ET.SubElement(frame, 'freejoint', dict(name='walker'))
  frame.append(walker_root)
  # position actuators scaled to ctrl in [-1, 1] over the joint range (scaled_actuators.py:71-81)
  joints = {j.get('name'): j for j in root.iter('joint')}
  dflt = root.find('default')
  gen = dflt.find('general')
  gen.set('forcelimited', 'true')
  act = root.find('actuator')
  for e in list(act):
    act.remove(e)
  for name, flo, fhi, kp in position_actuators():
    lo, hi = [float(x) for x in joints[name].get('range').split()]
    slope = (hi - lo) / 2.0
    g0, b0, b1 = kp * slope, kp * (lo + slope), -kp
    ET.SubElement(act, 'general', dict(name=name, joint=name, biastype='affine', gainprm=repr(g0),
                                       biasprm='%r %r 0' % (b0, b1), ctrllimited='true', ctrlrange='-1 1',
                                       forcerange='%r %r' % (flo, fhi)))
  out = ['<!-- Physics of BASELINE config 4: CMU humanoid (2019, position-controlled) on the composer Floor arena.',
         '     Restated from locomotion/walkers/assets/humanoid_CMU_V2019.xml, walkers/cmu_humanoid.py:53-110,360-399,',
         '     walkers/scaled_actuators.py:71-81, arenas/floors.py:76-82, composer/arena.xml:2-4 by',
         '     scripts/make_cmu_floor_model.py; rendering-only elements are omitted. -->']
  restate_model.dump(root, out)
  open(dst, 'w').write('\n'.join(out) + '\n')


if __name__ == '__main__':
  main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             'dm_control_amd/suite/assets/cmu_2019_position_floor.xml'))
