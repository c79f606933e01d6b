"""Writes the physics-only restatement of a reference suite model into suite/assets/.

Drops everything that never reaches the physics step (includes of visual / skybox / material
files, <asset>, <visual>, <statistic>, lights, cameras, material / rgba / group / width attributes)
and re-serialises what is left.  Only runnable where the reference tree exists (this container);
`tests/test_compiler.py` style checks then compare the compiled blobs of both files."""
import sys
import xml.etree.ElementTree as ET

DROP_ELEMS = {'include', 'asset', 'visual', 'statistic', 'light', 'camera'}
DROP_ATTRS = {'material', 'rgba', 'group', 'width', 'mark', 'markrgb'}


def strip(e):
  for c in list(e):
    if c.tag in DROP_ELEMS:
      e.remove(c)
    else:
      strip(c)
  for a in list(e.attrib):
    if a in DROP_ATTRS:
      del e.attrib[a]
  # a <default>/<tendon>/<site> element that only carried rendering attributes
  for c in list(e):
    if e.tag == 'default' and c.tag != 'default' and not c.attrib and not len(c):
      e.remove(c)


def dump(e, out, ind=0):
  pad = '  ' * ind
  attrs = ''.join(' %s="%s"' % (k, ' '.join(v.split())) for k, v in e.attrib.items())
  kids = list(e)
  if kids:
    out.append('%s<%s%s>' % (pad, e.tag, attrs))
    for c in kids:
      dump(c, out, ind + 1)
    out.append('%s</%s>' % (pad, e.tag))
  else:
    out.append('%s<%s%s />' % (pad, e.tag, attrs))


def main(src, dst, name):
  root = ET.fromstring(open(src).read())
  strip(root)
  out = ['<!-- Physics-only restatement of the suite "%s" model (reference: dm_control/suite/%s.xml).' % (name, name),
         '     Rendering-only elements (lights, cameras, materials, textures, visual / skybox includes) are omitted;',
         '     every number that reaches the physics step is unchanged. -->']
  dump(root, out)
  open(dst, 'w').write('\n'.join(out) + '\n')


if __name__ == '__main__':
  main(*sys.argv[1:4])
