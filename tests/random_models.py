"""Random MJCF models for parity fuzzing (test infrastructure): small articulated trees over a plane
with a random mix of the features both implementations support."""
import numpy as np


def random_model_xml(seed, ellipsoids=False, noslip=0, cylinders=False):
  rs = np.random.RandomState(seed)
  rs2 = np.random.RandomState(7919 + seed)   # separate stream: the base models keep their seeds
  cone = rs.choice(['pyramidal', 'elliptic'])
  integrator = rs.choice(['Euler', 'RK4'], p=[.75, .25])
  condim = int(rs.choice([1, 3, 3, 4, 6]))
  impratio = float(rs.choice([1.0, 1.0, 3.0]))
  density = float(rs.choice([0.0, 0.0, 800.0]))
  nroots = int(rs.choice([1, 1, 2]))
  out = ['<mujoco model="fuzz%d">' % seed,
         '<option timestep="%g" cone="%s" integrator="%s" impratio="%g" density="%g"/>' % (
             float(rs.choice([0.002, 0.005])), cone, integrator, impratio, density),
         '<default><geom condim="%d" friction="%g 0.01 0.002"/><joint damping="%g" armature="0.01"/></default>' % (
             condim, rs.uniform(.3, 1.2), rs.uniform(0.02, .5)),
         '<worldbody>', '<geom name="floor" type="plane" size="5 5 .1"/>']
  joints, sites, bodies, roots = [], [], [], []
  counter = [0]

  def body(depth, parent_len):
    k = counter[0]
    counter[0] += 1
    name = 'b%d' % k
    bodies.append(name)
    length = rs.uniform(.12, .3)
    radius = rs.uniform(.02, .05)
    lines = []
    if depth == 0:
      root_pos = (rs.uniform(-1, 1), rs.uniform(-1, 1), rs.uniform(.25, .6))
      roots.append(root_pos)
      lines.append('<body name="%s" pos="%g %g %g">' % ((name,) + root_pos))
      kind = rs.choice(['free', 'planar', 'hinge'])
      if kind == 'free':
        lines.append('<freejoint name="j%d"/>' % k)
      elif kind == 'planar':
        for ax, nm in (('1 0 0', 'x'), ('0 0 1', 'z')):
          lines.append('<joint name="j%d%s" type="slide" axis="%s"/>' % (k, nm, ax))
          joints.append('j%d%s' % (k, nm))
        lines.append('<joint name="j%dr" type="hinge" axis="0 1 0"/>' % k)
        joints.append('j%dr' % k)
      else:
        lines.append('<joint name="j%d" type="hinge" axis="0 1 0"/>' % k)
        joints.append('j%d' % k)
    else:
      # off-plane offsets: coplanar capsule chains cross with their axes intersecting exactly, where the
      # contact normal is undefined (dist/0) and any two implementations disagree
      lines.append('<body name="%s" pos="%g %g %g">' % (name, parent_len, rs.choice([-1, 1]) * rs.uniform(.03, .09),
                                                         rs.uniform(-.03, .03)))
      kind = rs.choice(['hinge', 'hinge', 'slide', 'ball', 'two'])
      if kind == 'ball':
        lines.append('<joint name="j%d" type="ball"/>' % k)
      elif kind == 'two':
        lines.append('<joint name="j%da" type="hinge" axis="0 1 0" range="-50 50" limited="true"/>' % k)
        lines.append('<joint name="j%db" type="hinge" axis="1 0 0" stiffness="%g"/>' % (k, rs.uniform(0, 2)))
        joints.extend(['j%da' % k, 'j%db' % k])
      elif kind == 'slide':
        lines.append('<joint name="j%d" type="slide" axis="1 0 0" range="-.05 .08" limited="true"/>' % k)
        joints.append('j%d' % k)
      else:
        ax = rs.choice(['0 1 0', '0 0 1', '1 1 0'])
        lim = rs.rand() < .6
        fl = ' frictionloss="%g"' % rs.uniform(.01, .2) if rs.rand() < .25 else ''
        lines.append('<joint name="j%d" type="hinge" axis="%s"%s%s/>' % (k, ax, ' range="-70 70" limited="true"' if lim else '', fl))
        joints.append('j%d' % k)
    if rs.rand() < .7:
      lines.append('<geom name="g%d" type="capsule" fromto="0 0 0 %g %g %g" size="%g"/>' % (
          k, length, rs.uniform(-.04, .04), rs.uniform(-.04, .04), radius))
    else:
      lines.append('<geom name="g%d" type="sphere" pos="%g 0 0" size="%g"/>' % (k, length / 2, radius * 1.5))
    if ellipsoids and rs2.rand() < .6:
      # an ellipsoid "hand" past the end of the link (ellipsoid-plane / -sphere / -capsule / -ellipsoid pairs)
      q = rs2.normal(size=4)
      lines.append('<geom name="e%d" type="ellipsoid" pos="%g %g 0" quat="%g %g %g %g" size="%g %g %g"/>' % (
          (k, length + .03, rs2.uniform(-.03, .03)) + tuple(q / np.linalg.norm(q)) + tuple(rs2.uniform(.02, .07, 3))))
    lines.append('<site name="s%d" pos="%g 0 0" size=".01"/>' % (k, length / 2))
    sites.append('s%d' % k)
    nchild = 0 if depth >= 3 else int(rs.choice([0, 1, 1, 2]))
    if counter[0] > 6:
      nchild = 0
    for _ in range(nchild):
      lines += body(depth + 1, length)
    lines.append('</body>')
    return lines

  for _ in range(nroots):
    out += body(0, 0.0)
  if cylinders:
    # static cylinders lying and standing where the trees come down (sphere-cylinder / capsule-cylinder pairs; not with
    # ellipsoids: that pair type is refused)
    rs3 = np.random.RandomState(104729 + seed)
    for c in range(3):
      q = rs3.randn(4)
      x0, y0, _ = roots[c % len(roots)]
      out.append('<geom name="cyl%d" type="cylinder" pos="%g %g %g" quat="%g %g %g %g" size="%g %g"/>' % (
          (c, x0 + rs3.uniform(-.1, .4), y0 + rs3.uniform(-.1, .1), rs3.uniform(.02, .12)) + tuple(q / np.linalg.norm(q)) +
          (rs3.uniform(.05, .15), rs3.uniform(.03, .12))))
  out.append('</worldbody>')
  tendons, eqs = [], []
  hinges = [j for j in joints]
  if len(hinges) >= 2 and rs.rand() < .5:
    j1, j2 = rs.choice(hinges, 2, replace=False)
    tendons.append('<fixed name="tf"%s><joint joint="%s" coef="%g"/><joint joint="%s" coef="%g"/></fixed>' % (
        ' stiffness="%g" damping="%g"' % (rs.uniform(0, 3), rs.uniform(0, .2)) if rs.rand() < .5 else '',
        j1, rs.uniform(.3, 1), j2, -rs.uniform(.3, 1)))
    if rs.rand() < .3:
      eqs.append('<tendon tendon1="tf" solref=".01 1"/>')
  if len(sites) >= 2 and rs.rand() < .4:
    tendons.append('<spatial name="ts" limited="true" range="0 %g"><site site="%s"/><site site="%s"/></spatial>' % (
        rs.uniform(.3, .8), sites[0], sites[-1]))
  if tendons:
    out += ['<tendon>'] + tendons + ['</tendon>']
  if eqs:
    out += ['<equality>'] + eqs + ['</equality>']
  if joints:
    acts = []
    for j in joints:
      if rs.rand() < .6:
        if rs.rand() < .7:
          acts.append('<motor name="a_%s" joint="%s" gear="%g" ctrllimited="true" ctrlrange="-1 1"/>' % (j, j, rs.uniform(.5, 5)))
        else:
          acts.append('<position name="a_%s" joint="%s" kp="%g" ctrllimited="true" ctrlrange="-1 1"/>' % (j, j, rs.uniform(1, 10)))
    if not acts:
      acts.append('<motor name="a0" joint="%s" gear="1"/>' % joints[0])
    if any('name="tf"' in t for t in tendons) and not eqs:
      acts.append('<motor name="a_tf" tendon="tf" gear="%g"/>' % rs.uniform(.5, 3))
    out += ['<actuator>'] + acts + ['</actuator>']
  sens = []
  for s in sites:
    r = rs.rand()
    if r < .2:
      sens.append('<velocimeter site="%s"/>' % s)
    elif r < .35:
      sens.append('<gyro site="%s"/>' % s)
    elif r < .5:
      sens.append('<accelerometer site="%s"/>' % s)
    elif r < .6:
      sens.append('<touch site="%s"/>' % s)
    elif r < .7:
      sens.append('<framepos objtype="site" objname="%s"/>' % s)
    elif r < .78:
      sens.append('<force site="%s"/>' % s)
    elif r < .84:
      sens.append('<torque site="%s"/>' % s)
    elif r < .9:
      sens.append('<frame%saxis objtype="site" objname="%s"/>' % (rs.choice(['x', 'y', 'z']), s))
    elif r < .925:
      sens.append('<framequat objtype="site" objname="%s"/>' % s)
    elif r < .95:
      sens.append('<framelinvel objtype="site" objname="%s"/>' % s)
    elif r < .975:
      sens.append('<frameangvel objtype="site" objname="%s"/>' % s)
    else:
      sens.append('<rangefinder site="%s"/>' % s)
  sens.append('<subtreelinvel body="%s"/>' % bodies[0])
  sens.append('<subtreecom body="%s"/>' % bodies[-1])
  if joints:
    sens.append('<jointpos joint="%s"/>' % joints[0])
    sens.append('<jointvel joint="%s"/>' % joints[-1])
  out += ['<sensor>'] + sens + ['</sensor>', '</mujoco>']
  xml = '\n'.join(out)
  if noslip:
    xml = xml.replace('<option ', '<option noslip_iterations="%d" ' % noslip, 1)
  return xml
