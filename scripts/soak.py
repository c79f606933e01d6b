"""Soak: every suite task, batch of 256 environments, fp32, random actions for a few thousand env-steps
through the host Environment API; reports throughput, reward range, warnings."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dm_control_amd import suite

B = 256
T = int(os.environ.get('T', 300))
out = []
_ONLY = os.environ.get('DOMAINS')
for domain, task in [dt for dt in suite.ALL_TASKS if not _ONLY or dt[0] in _ONLY.split(',')]:
  try:
    env = suite.load(domain, task, task_kwargs=dict(random=0), physics_kwargs=dict(batch_size=B, precision=32))
    spec = env.action_spec()
    rs = np.random.RandomState(0)
    # unbounded action specs (lqr: no ctrlrange, engine.py:1093-1103 gives +-mjMAXVAL) are sampled in [-1, 1]: U(-1e10, 1e10)
    # forces drive the state past mjMAXVAL within a few hundred steps and raise BADQACC by construction
    lo, hi = np.maximum(spec.minimum, -1.0), np.minimum(spec.maximum, 1.0)
    ts = env.reset()
    rmin, rmax, n = np.inf, -np.inf, 0
    t0 = time.perf_counter()
    with env.physics.suppress_physics_errors():
      for t in range(T):
        ts = env.step(rs.uniform(lo, hi, spec.shape))
        n += 1
        if ts.reward is not None:
          r = np.asarray(ts.reward); rmin = min(rmin, float(r.min())); rmax = max(rmax, float(r.max()))
        if ts.last():
          ts = env.reset()
    dt = time.perf_counter() - t0
    obs_ok = all(np.all(np.isfinite(np.asarray(v))) for v in ts.observation.values())
    w = env.physics.batch.get('warning').sum(axis=0).tolist()
    r = dict(task='%s/%s' % (domain, task), steps=n, env_steps_per_s=B * n / dt, reward=[rmin, rmax], obs_finite=bool(obs_ok), warnings=w)
    env.physics.free()
  except Exception as ex:  # pylint: disable=broad-except
    r = dict(task='%s/%s' % (domain, task), error=repr(ex)[:200])
  print(json.dumps(r), flush=True)
  out.append(r)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'soak.json'), 'w'), indent=1)
