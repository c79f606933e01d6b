"""Builds, without a GPU, what suite/fused_env.py compiles at environment creation for every suite task: the generated
task header, its stand-alone kernels and the step kernel specialised for (model, task) with the task layer as its
epilogue -- into the in-tree cache that travels with the checkout (dm_control_amd/_spec_cache), so that a fresh GPU box
starts from a warm cache.  The trace runs on the CPU oracle stand-in (test infrastructure) at batch size 2: the generated
code does not depend on the batch size.   python scripts/prebuild_task_kernels.py [precision] [domain-task ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def work(arg):
  domain, task, precision = arg
  import oracle_backend as ob
  from dm_control_amd import physics
  physics.BatchedPhysics = ob.OracleBatch
  from dm_control_amd import suite, specialise
  from dm_control_amd.suite import common, fused_env
  t0 = time.time()
  env = suite.load(domain, task, task_kwargs=dict(random=0), physics_kwargs=dict(batch_size=2, precision=64))
  env.reset()
  prog = fused_env.trace(env, precision=precision, title='%s.%s' % (domain, task))
  prog.build()
  m = env.physics.model
  kw = common.physics_kwargs(domain, {})
  caps = (int(kw.get('nconmax', 0)), int(kw.get('njmax', 0)), int(kw.get('njcon', 0)))
  lpe = 32 if m.nv <= 12 else 64
  p = specialise.build(m, precision, lpe, caps, task_header=prog.header_path)
  return '%s %s: %s (%.0f s)' % (domain, task, os.path.basename(p), time.time() - t0)


if __name__ == '__main__':
  import multiprocessing as mp
  from dm_control_amd import suite
  precision = int(sys.argv[1]) if len(sys.argv) > 1 else 32
  only = sys.argv[2:]
  tasks = [(d, t, precision) for d, t in sorted(suite.ALL_TASKS) if not only or '%s-%s' % (d, t) in only]
  with mp.get_context('spawn').Pool(int(os.environ.get('JOBS', 6))) as pool:
    for line in pool.imap_unordered(work, tasks):
      print(line, flush=True)
