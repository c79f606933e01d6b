"""numpy restatement of dmc_batch_randomize_joints (include/dmc_batch.h) -- TEST INFRASTRUCTURE: the GPU tests compare
the device kernel's draws with it, the CPU tests pin its Philox4x32-10 on the published known-answer vectors
(Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 kat_vectors)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF
LIMITED, UNLIMITED_HINGE, QUATERNION, FREE_NORMAL, ALL = 1, 2, 4, 8, 7
FREE, BALL, SLIDE, HINGE = 0, 1, 2, 3


def philox4x32_10(ctr, key):
  c0, c1, c2, c3 = [int(c) & MASK for c in ctr]
  k0, k1 = [int(k) & MASK for k in key]
  for _ in range(10):
    p0, p1 = M0 * c0, M1 * c2
    c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
    k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
  return c0, c1, c2, c3


def u01(x):
  return (float(x) + 0.5) / 4294967296.0


def box_muller(u0, u1):
  r, a = np.sqrt(-2.0 * np.log(u0)), 2 * np.pi * u1
  return r * np.cos(a), r * np.sin(a)


def randomize_joints(model, qpos, seed, draw, env_mask=None, flags=ALL):
  """qpos: (B, nq) array edited in place; draw: (B,) int counters, incremented for the environments drawn."""
  B = qpos.shape[0]
  for env in range(B):
    if env_mask is not None and not env_mask[env]:
      continue
    k = int(draw[env]); draw[env] = k + 1
    key = (seed & MASK, (seed >> 32) & MASK)     # the seed alone; env is the fourth counter word
    for j in range(model.njnt):
      t, a, lim = int(model.jnt_type[j]), int(model.jnt_qposadr[j]), int(model.jnt_limited[j])
      lo, hi = model.jnt_range[j]
      x = philox4x32_10((k, j, 0, env), key)
      if t in (HINGE, SLIDE):
        if lim:
          if flags & LIMITED:
            qpos[env, a] = lo + (hi - lo) * u01(x[0])
        elif t == HINGE and flags & UNLIMITED_HINGE:
          qpos[env, a] = -np.pi + 2 * np.pi * u01(x[0])
      elif t == BALL and lim:
        if not flags & LIMITED:
          continue
        y = philox4x32_10((k, j, 1, env), key)
        n = np.array(box_muller(u01(x[0]), u01(x[1])) + box_muller(u01(x[2]), u01(x[3])))[:3]
        ang = u01(y[0]) * hi
        qpos[env, a] = np.cos(0.5 * ang)
        qpos[env, a + 1:a + 4] = n * np.sin(0.5 * ang) / np.linalg.norm(n)
      elif t in (BALL, FREE):
        if not flags & QUATERNION:
          continue
        if t == BALL or flags & FREE_NORMAL:
          q = np.array(box_muller(u01(x[0]), u01(x[1])) + box_muller(u01(x[2]), u01(x[3])))
        else:
          q = np.array([u01(c) for c in x])
        a0 = a + (3 if t == FREE else 0)
        qpos[env, a0:a0 + 4] = q / np.linalg.norm(q)
  return qpos
