"""Extracts the MuJoCo-generated kinematics the reference holds into tests/golden/cmu2019_mocap.json.

Source (read-only, this container only): dm_control/locomotion/mocap/test_00{1,2}.textproto -- two 10-frame
clips of the CMU 2019 walker.  Schema: locomotion/mocap/mocap.proto:105-146 (WalkerPose); the features were
written by locomotion/tasks/reference_pose/utils.py:127-160 `get_features` from a real-MuJoCo `Physics`:
  position / quaternion  = walker root pose            joints = qpos of walkers/cmu_humanoid.py:36-50 in that order
  body_positions / body_quaternions = data.xpos / data.xquat of `mocap_tracking_bodies` (every body but `root`,
                                      document order; cmu_humanoid.py:331-336)
  end_effectors / appendages = egocentric positions of (rradius, lradius, rfoot, lfoot[, head])
                               (cmu_humanoid.py:315-319,473-482)
Run:  python scripts/make_mocap_golden.py   (needs /root/reference; the fixture it writes is committed)
"""
import json
import os
import re
import sys

REF = '/root/reference/dm_control/locomotion'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'cmu2019_mocap.json')
FIELDS = ('position', 'quaternion', 'joints', 'end_effectors', 'appendages', 'body_positions', 'body_quaternions',
          'velocity', 'angular_velocity', 'joints_velocity', 'center_of_mass')


def parse_textproto(text):
  """Minimal protobuf text-format reader: nested `name { ... }` blocks, `key: value`, `key: [v, v, ...]`.
  Repeated keys accumulate into lists."""
  tok = re.compile(r'\s*(?:(\w+)\s*\{|(\})|(\w+)\s*:\s*(\[[^\]]*\]|"[^"]*"|[^\s]+))')
  root = {}
  stack = [root]
  pos = 0
  while True:
    m = tok.match(text, pos)
    if not m:
      if text[pos:].strip():
        raise ValueError('unparsed text at %d: %r' % (pos, text[pos:pos + 40]))
      break
    pos = m.end()
    if m.group(1):
      child = {}
      stack[-1].setdefault(m.group(1), []).append(child)
      stack.append(child)
    elif m.group(2):
      stack.pop()
    else:
      key, val = m.group(3), m.group(4)
      if val.startswith('['):
        val = [float(v) for v in val[1:-1].split(',') if v.strip()]
      elif val.startswith('"'):
        val = val[1:-1]
      else:
        try:
          val = float(val)
        except ValueError:
          pass
      stack[-1].setdefault(key, []).append(val)
  return root


def mocap_joint_order():
  src = open(os.path.join(REF, 'walkers', 'cmu_humanoid.py')).read()
  block = re.search(r'_CMU_MOCAP_JOINTS = \((.*?)\)', src, re.S).group(1)
  return re.findall(r"'(\w+)'", block)


def main():
  clips = []
  for name in ('test_001.textproto', 'test_002.textproto'):
    msg = parse_textproto(open(os.path.join(REF, 'mocap', name)).read())
    frames = []
    for ts in msg['timesteps']:
      w = ts['walkers'][0]
      frames.append({k: w[k][0] for k in FIELDS if k in w})
    clips.append({'identifier': msg['identifier'][0], 'dt': msg['dt'][0], 'source': 'dm_control/locomotion/mocap/' + name,
                  'frames': frames})
  out = {
      'comment': 'MuJoCo-generated kinematics of the CMU 2019 walker held by the reference; written by '
                 'scripts/make_mocap_golden.py, do not edit',
      'joint_order': mocap_joint_order(),
      'end_effector_bodies': ['rradius', 'lradius', 'rfoot', 'lfoot'],
      'appendage_bodies': ['rradius', 'lradius', 'rfoot', 'lfoot', 'head'],
      'tracking_bodies': 'every body of the walker except root, document order',
      'clips': clips,
  }
  with open(OUT, 'w') as f:
    json.dump(out, f, separators=(',', ':'))
  n = sum(len(c['frames']) for c in clips)
  print('wrote %s: %d clips, %d frames, %d bytes' % (OUT, len(clips), n, os.path.getsize(OUT)), file=sys.stderr)


if __name__ == '__main__':
  main()
