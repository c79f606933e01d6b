"""Writes suite/assets/soccer_2v2_boxhead.xml: the physics of BASELINE config 5.

The reference composes that model at run time with PyMJCF (locomotion/soccer/__init__.py:92-148): the
composer arena (composer/arena.xml: elliptic cones, 5 noslip sweeps, radians), a RandomizedPitch
(soccer/pitch.py: ground plane, four wall planes, two goals of ten capsule posts each), four BoxHead
walkers (soccer/boxhead.py + assets/boxhead/boxhead.xml: three root slides, steer hinge, kick slider,
rolling ball; camera joints removed, roll gear -60), and the SoccerBall (soccer/soccer_ball.py: condim 6,
priority 1), physics timestep 0.005 (soccer/task.py:105-106).  PyMJCF is not importable here, so this script
restates the composition from those sources.  The pitch size is randomised per episode in the reference
(32 x 24 ... 48 x 36 half-extents); this file uses the midpoint 40 x 30.  Rendering-only elements, the
non-colliding perimeter planes, hoarding and detector sites are omitted; the players stand where a
kick-off would put them.  Only runnable where the reference tree exists."""
import copy
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import restate_model

REF = '/root/reference/dm_control/locomotion/soccer'
SIZE = (40.0, 30.0)
SIDE_WIDTH = 32. / 6.
GOAL_LENGTH_RATIO = 0.33
GOALPOST_RELATIVE_SIZE = 0.07
SUPPORT_POST_RATIO = 0.75
GOALPOSTS = {'right_post': (1, -1, -1, 1, -1, 1), 'left_post': (1, 1, -1, 1, 1, 1), 'top_post': (1, -1, 1, 1, 1, 1),
             'right_base': (1, -1, -1, -1, -1, -1), 'left_base': (1, 1, -1, -1, 1, -1), 'back_base': (-1, -1, -1, -1, 1, -1),
             'right_support': (-1, -1, -1, .2, -1, 1), 'right_top_support': (.2, -1, 1, 1, -1, 1),
             'left_support': (-1, 1, -1, .2, 1, 1), 'left_top_support': (.2, 1, 1, 1, 1, 1)}
BUILTIN_GEOM = dict(condim='3', friction='1 0.005 0.0001', solref='0.02 1', solimp='0.9 0.95 0.001 0.5 2')


def fmt(v):
  return ' '.join(repr(float(x)) for x in v)


def main(dst):
  root = ET.Element('mujoco', dict(model='soccer_2v2_boxhead'))
  ET.SubElement(root, 'compiler', dict(coordinate='local', angle='radian', eulerseq='xyz', boundmass='1e-5', boundinertia='1e-11'))
  ET.SubElement(root, 'option', dict(cone='elliptic', noslip_iterations='5', noslip_tolerance='0', timestep='0.005'))
  box = ET.fromstring(open(os.path.join(REF, 'assets/boxhead/boxhead.xml')).read())
  restate_model.strip(box)
  # PyMJCF scopes the walker's defaults to the walker: here they become the class "boxhead"
  dflt = ET.SubElement(root, 'default')
  scoped = ET.SubElement(dflt, 'default', {'class': 'boxhead'})
  for c in box.find('default'):
    if c.tag != 'mesh':
      scoped.append(c)
  wb = ET.SubElement(root, 'worldbody')
  # pitch.py:403-422: ground and walls (planes collide as infinite half-spaces whatever their size)
  ET.SubElement(wb, 'geom', dict(name='ground', type='plane', size=fmt(SIZE + (max(SIZE) / 100,)), **BUILTIN_GEOM))
  walls = [((0., -SIZE[1], 0.), (-1, 0, 0, 0, 0, 1)), ((0., SIZE[1], 0.), (1, 0, 0, 0, 0, 1)),
           ((-SIZE[0], 0., 0.), (0, 1, 0, 0, 0, 1)), ((SIZE[0], 0., 0.), (0, -1, 0, 0, 0, 1))]
  for k, (pos, xyaxes) in enumerate(walls):
    ET.SubElement(wb, 'geom', dict(name='wall%d' % k, type='plane', pos=fmt(pos), xyaxes=fmt(xyaxes), size='1e-7 1e-7 1e-7', **BUILTIN_GEOM))
  # pitch.py:165-283, 426-446: goals
  goal_size = (SIDE_WIDTH / 2, SIZE[1] * GOAL_LENGTH_RATIO, SIDE_WIDTH / 2)
  radius = GOALPOST_RELATIVE_SIZE * sum(goal_size) / 3.
  for gname, direction, pos in (('home_goal', 1, (-SIZE[0] + goal_size[0], 0, goal_size[2])),
                                ('away_goal', -1, (SIZE[0] - goal_size[0], 0, goal_size[2]))):
    d3 = np.array((direction, direction, 1.))
    for pname, unit in GOALPOSTS.items():
      fromto = np.array(unit) * np.hstack((d3, d3)) * np.array(goal_size + goal_size) + np.array(pos + pos)
      r = radius * (1.01 if 'top' in pname else 1.0) * (SUPPORT_POST_RATIO if 'support' in pname else 1.0)
      ET.SubElement(wb, 'geom', dict(name='%s/%s' % (gname, pname), type='capsule', size=repr(float(r)), fromto=fmt(fromto), **BUILTIN_GEOM))
  # players: attachment frame with the three root slides (boxhead.py:277-285), then the walker's worldbody
  sensors = ET.SubElement(root, 'sensor')
  actuators = ET.SubElement(root, 'actuator')
  spots = {'home0': (-10, 5), 'home1': (-10, -5), 'away0': (10, 5), 'away1': (10, -5)}
  for pname, (x, y) in spots.items():
    w = copy.deepcopy(box)
    pre = pname + '/'
    frame = ET.SubElement(wb, 'body', dict(name=pre[:-1], pos=fmt((x, y, 0)), childclass='boxhead'))
    for ax, axis in (('x', '1 0 0'), ('y', '0 1 0'), ('z', '0 0 1')):
      ET.SubElement(frame, 'joint', {'name': pre + 'root_' + ax, 'type': 'slide', 'axis': axis, 'class': 'root'})
    for e in w.find('worldbody'):
      frame.append(e)
    # boxhead.py:236-240: no camera control -> the camera joints and their actuators are removed
    for parent in frame.iter():
      for c in list(parent):
        if c.tag == 'joint' and c.get('name') in ('camera_yaw', 'camera_pitch'):
          parent.remove(c)
    for e in frame.iter():
      if e is not frame and 'name' in e.attrib and not e.get('name').startswith(pre):
        e.set('name', pre + e.get('name'))
    for a in w.find('actuator'):
      if a.get('name') in ('camera_yaw', 'camera_pitch'):
        continue
      a.set('joint', pre + a.get('joint'))
      if a.get('name') == 'roll':
        a.set('gear', '-60')        # boxhead.py:164 roll_gear, :241-242
      a.set('name', pre + a.get('name'))
      a.set('class', 'boxhead')
      actuators.append(a)
    for sn in w.find('sensor'):
      sn.set('site', pre + sn.get('site'))
      sn.set('name', pre + sn.get('name'))
      sensors.append(sn)
  # ball: soccer_ball.py:48-95 attached through a free joint, dropped from _INIT_BALL_Z = 0.5 (initializers.py:22)
  ball = ET.SubElement(wb, 'body', dict(name='soccer_ball', pos='0 0 0.5'))
  ET.SubElement(ball, 'freejoint', dict(name='soccer_ball'))
  ET.SubElement(ball, 'geom', dict(name='soccer_ball/geom', type='sphere', pos='0 0 0.35', size='0.35', condim='6', priority='1',
                                   mass='0.045', friction='0.7 0.075 0.075', solref='0.02 1.0', solimp='0.9 0.95 0.001 0.5 2'))
  for tag, sname in (('framepos', 'position'), ('framequat', 'orientation'), ('framelinvel', 'linear_velocity'), ('frameangvel', 'angular_velocity')):
    ET.SubElement(sensors, tag, dict(name='soccer_ball/' + sname, objtype='geom', objname='soccer_ball/geom'))
  out = ['<!-- Physics of BASELINE config 5: soccer 2v2 with BoxHead walkers on a 40 x 30 pitch.',
         '     Restated from locomotion/soccer/{__init__,pitch,boxhead,soccer_ball,task}.py, assets/boxhead/boxhead.xml and',
         '     composer/arena.xml by scripts/make_soccer_model.py; rendering-only and non-colliding elements are omitted. -->']
  restate_model.dump(root, out)
  open(dst, 'w').write('\n'.join(out) + '\n')


if __name__ == '__main__':
  main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             'dm_control_amd/suite/assets/soccer_2v2_boxhead.xml'))
