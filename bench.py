#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched physics step (BASELINE.json metric).

    python bench.py [--config {2,3,4,5}] [--gpus N] [--steps K] [--warmup W]

Workloads = BASELINE.json configs; the default is config 2, the one the metric is quoted on:

  2  suite 'cheetah run', batch 4096, fp32, 1 physics step per env-step (suite/cheetah.py:48); start states as
     Cheetah.initialize_episode (suite/cheetah.py:63-76): limited joints ~ U(range), 200 settle steps, time = 0
  3  suite 'humanoid stand', batch 4096, 5 physics steps per env-step (suite/humanoid.py:30,55); start states by
     rejection until ncon == 0 (suite/humanoid.py:160-165)
  4  locomotion CMU humanoid (2019, position-controlled) on the Floor arena, 6 physics steps per env-step
     (locomotion/examples/basic_cmu_2019.py:110-111), batch 4096 PER GPU (32768 over 8); upright start pose at a
     uniform position of the arena (tasks/go_to_target.py:139-165)
  5  locomotion.soccer 2v2 BoxHead, 5 physics steps per env-step (soccer/task.py:105-106), batch 256 PER GPU
     (2048 over 8); players and ball spread over the pitch (soccer/initializers.py)

Actions are U(-1, 1)^nu, generated once and resident in HBM.  One "step" = one `Physics.step(n_sub_steps)` over the
whole batch = ONE launch of the fused HIP kernel with legacy_step semantics (engine.py:147-162).

N > 1: one process per GPU, environments sharded in contiguous ranges (dm_control_amd/sharding.py); the physics has
no data-path collective, `value` is measured with every rank stepping its shard from actions resident on its own
GPU (weak scaling).  The agent-interface exchange (actions scattered from rank 0, observations gathered to every
rank over RCCL) is timed as a separate leg and reported under "collectives".

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes this file under `torch.distributed.run` with N
ranks on 127.0.0.1 (one GPU each); under a launcher the world size must equal --gpus or the run fails.  On a box with
fewer than N GPUs the run fails too, unless DMC_BENCH_SINGLE_DEVICE=1 (+ DMC_BENCH_BACKEND=gloo) asks for all ranks on
cuda:0 -- a harness check, not a measurement.

roofline.traffic and roofline_issue are measured IN THE RUN when rocprofv3 is on the PATH (N = 1): the bench re-runs
itself (`--pmc-child`: same workload, 30 steps, nothing else) under `rocprofv3 --kernel-trace --pmc <counters>` once per
counter group -- FETCH_SIZE and WRITE_SIZE in separate passes, reads x 2 as MI355X_MICROARCH.md prescribes for gfx950
-- and reads the timed launches' rows from the csv.  DMC_BENCH_NO_PMC=1 (or an enclosing rocprofv3) skips the passes;
the line then carries the committed counters of profiles/ and says so in `traffic_source`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec
MAX_CLOCK_GHZ = 2.4                # MI355X_MICROARCH.md: peak engine clock
SIMDS = 1024                       # 256 CUs x 4 SIMDs
# issue cost of one wave64 fp32 VALU instruction on a CDNA4 SIMD: 2 cycles (MI355X_MICROARCH.md "Wave scheduling" and the
# measured v_fma_f32 row of "Per-instruction cycle constants": 157 TF fp32 = 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz).
# The SQ counters tick in quad-cycles (same table, s_memtime row); round 2 priced an instruction at one quad-cycle = 4.
VALU_CYCLES_PER_INST = 2
PMC_PASSES = (('FETCH_SIZE',), ('WRITE_SIZE',),
              ('SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'))
METRIC = 'env-steps/s (whole node) at batch 4096; max rel qpos error vs CPU'

# algo_bytes: SURVEY.md 8(d) compulsory bytes per env per launch, fp32 SoA with the state kept on chip across the
# substeps: 4 [ (nq + nv + nv_warm + nu) read + (nq + nv + nv_warm + nsensordata) written ] + 8 (time)
CONFIGS = {
    2: dict(asset='cheetah', nsub=1, batch=4096, algo_bytes=260, outputs=('sensor',), parity_envs=64, parity_steps=1000,
            workload="suite 'cheetah run'"),
    3: dict(asset='humanoid', nsub=5, batch=4096, algo_bytes=1012, outputs=('sensor', 'xpos', 'xmat', 'subtree_com'),
            parity_envs=64, parity_steps=100, workload="suite 'humanoid stand'"),
    4: dict(asset='cmu_2019_position_floor', nsub=6, batch=4096, algo_bytes=1828, outputs=('sensor', 'xpos', 'xmat'),
            parity_envs=64, parity_steps=50, workload='locomotion CMU humanoid (2019, position-controlled) on the Floor arena'),
    5: dict(asset='soccer_2v2_boxhead', nsub=5, batch=256, algo_bytes=928, outputs=('sensor', 'xpos', 'xmat'),
            parity_envs=64, parity_steps=100, workload='locomotion.soccer 2v2 BoxHead'),
}


def parse(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS))
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=None)
  ap.add_argument('--warmup', type=int, default=None)
  ap.add_argument('--batch', type=int, default=None, help='envs per GPU')
  ap.add_argument('--lanes', type=int, default=int(os.environ.get('DMC_LANES', '0')))
  ap.add_argument('--precision', type=int, default=32)
  ap.add_argument('--parity-envs', type=int, default=None)
  ap.add_argument('--parity-steps', type=int, default=None)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--pipeline', type=str, default='2,4', help='part-batch counts of the pipelined leg, comma-separated (0 / 1 = skip)')
  ap.add_argument('--cpu-envs', type=int, default=4096)
  ap.add_argument('--cpu-seconds', type=float, default=10.0)
  ap.add_argument('--pmc-child', action='store_true', help='internal: only the timed launches (run under rocprofv3 by the parent)')
  ap.add_argument('--extra', type=int, default=None,
                  help='short legs of the other BASELINE configs (3, 4, 5) and of Environment.step on config 2 under "extra" '
                       '(default: on for the default invocation -- config 2 on one GPU -- off otherwise)')
  ap.add_argument('--extra-steps', type=int, default=12, help='timed launches of each extra config leg (>= 10)')
  return ap.parse_args(argv)


def launch_command(args, argv, port):
  """The command `--gpus N` (N > 1, no launcher in the environment) re-executes: N ranks of this file on 127.0.0.1."""
  return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def check_world(args, world, n_devices, single_device):
  """--gpus is a contract, not a hint: the world must be exactly N ranks and (outside the single-device harness
  mode) the box must have N GPUs."""
  if world != args.gpus:
    raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
  if not single_device and n_devices < args.gpus:
    raise SystemExit('bench.py: --gpus %d but only %d GPU(s) are visible (DMC_BENCH_SINGLE_DEVICE=1 with '
                     'DMC_BENCH_BACKEND=gloo runs all ranks on cuda:0 as a harness check)' % (args.gpus, n_devices))


def free_port():
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    return sk.getsockname()[1]


def collect_pmc(args, K):
  """HBM bytes and SQ counters per timed launch, measured now: this file re-run as `--pmc-child` under rocprofv3, one
  pass per counter group (separate --pmc passes, --kernel-trace only beside them).  Returns (dict, why-not)."""
  exe = shutil.which('rocprofv3')
  if os.environ.get('DMC_BENCH_NO_PMC'):
    return None, 'DMC_BENCH_NO_PMC set'
  if exe is None:
    return None, 'rocprofv3 not on PATH'
  if any(k.startswith('ROCPROF') or k.startswith('ROCP_') for k in os.environ):
    return None, 'already running under a profiler'
  import csv
  import glob
  import statistics
  out = {}
  child = [sys.executable, os.path.abspath(__file__), '--pmc-child', '--config', str(args.config), '--steps', str(K),
           '--warmup', '5', '--precision', str(args.precision), '--lanes', str(args.lanes)]
  if args.batch:
    child += ['--batch', str(args.batch)]
  env = dict(os.environ, TMPDIR='/tmp')
  for counters in PMC_PASSES:
    with tempfile.TemporaryDirectory(dir='/tmp') as td:
      cmd = [exe, '--kernel-trace', '--pmc'] + list(counters) + ['-d', td, '-o', 'p', '--output-format', 'csv', '--'] + child
      try:
        r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
      except subprocess.TimeoutExpired:
        return (out or None), 'rocprofv3 pass %s timed out' % (counters,)
      if r.returncode != 0:
        return (out or None), 'rocprofv3 pass %s rc=%d: %s' % (counters, r.returncode, r.stderr.decode()[-200:])
      dur = {}
      for f in glob.glob(os.path.join(td, '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
          dur[row.get('Dispatch_Id')] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
      per = {}
      for f in glob.glob(os.path.join(td, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
          if 'step_kernel' in row.get('Kernel_Name', ''):
            d = per.setdefault(int(row['Dispatch_Id']), {})
            d[row['Counter_Name']] = d.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
      ids = sorted(per)[-K:]                       # the child's last K step_kernel dispatches are the timed launches
      if len(ids) < K:
        return (out or None), 'rocprofv3 pass %s: %d step_kernel dispatches, expected >= %d' % (counters, len(ids), K)
      for c in counters:
        out[c] = statistics.median(per[i].get(c, 0.0) for i in ids)
      us = [dur[str(i)] for i in ids if str(i) in dur]
      if us:
        out['kernel_us_under_pmc:' + counters[0]] = statistics.median(us)
  return out, None


def load_model(asset):
  from dm_control_amd import mjcf_compiler
  from dm_control_amd.suite import common
  return mjcf_compiler.compile_xml(common.read_model(asset + '.xml'))


def initial_qpos(cfg, model, n, seed0, phys=None):
  """Per-env start configurations (host array (n, nq)); `phys` (a BatchedPhysics of n envs) evaluates the
  humanoid's contact-free rejection test for the whole batch at once."""
  m = model
  q = np.tile(m.qpos0, (n, 1))
  asset = cfg['asset']
  if asset == 'cheetah':
    lim = m.jnt_limited == 1
    lo, hi = m.jnt_range[lim].T
    for e in range(n):
      q[e, lim] = np.random.RandomState(seed0 + e).uniform(lo, hi)
  elif asset == 'humanoid':
    rs = np.random.RandomState(seed0)
    todo = np.ones(n, bool)
    for _ in range(200):
      k = int(todo.sum())
      if not k:
        break
      for j in range(m.njnt):      # randomize_limited_and_rotational_joints (suite/utils/randomizers.py:35-88)
        t, a = m.jnt_type[j], m.jnt_qposadr[j]
        if m.jnt_limited[j] and t in (2, 3):
          q[todo, a] = rs.uniform(m.jnt_range[j][0], m.jnt_range[j][1], k)
        elif t == 3:
          q[todo, a] = rs.uniform(-np.pi, np.pi, k)
        elif t == 0:
          quat = rs.randn(k, 4)
          q[todo, a + 3:a + 7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
      phys.set('qpos', q)
      phys.forward(disable_actuation=True)
      todo = phys.get('ncon')[:, 0] > 0
    else:
      raise RuntimeError('could not find collision-free humanoid start states')
  elif asset == 'cmu_2019_position_floor':
    rs = np.random.RandomState(seed0)
    q[:, 0:2] += rs.uniform(-4, 4, (n, 2))       # arena_position: U(-size/2, size/2), Floor size 8 x 8
  elif asset == 'soccer_2v2_boxhead':
    from dm_control_amd.composer.tasks import soccer
    rs = np.random.RandomState(seed0)
    q = np.tile(soccer.kickoff_qpos(m), (n, 1))      # PyMJCF's qpos0 has every entity at the origin
    adr = soccer.addresses(m)
    q[:, [a for xy in adr['players'] for a in xy]] += rs.uniform(-8, 8, (n, 8))      # players around their kick-off spots
    q[:, adr['ball_q']:adr['ball_q'] + 2] += rs.uniform(-15, 15, (n, 2))            # ball
  return q


def make_oracles(model, q, v=None, w=None, step1=True):
  from oracle import oracle
  om = oracle.OracleModel(model)
  out = []
  for e in range(q.shape[0]):
    p = oracle.OraclePhysics(om)
    p.qpos[:] = q[e]
    if v is not None:
      p.qvel[:] = v[e]
    if w is not None:
      p.qacc_warmstart[:] = w[e]
    if step1:
      p.step1()
    out.append(p)
  return out


def threaded_rollout(phys, acts, nsub, nthreads):
  """oracle.rollout_legacy over `phys` split across host threads (ctypes releases the GIL)."""
  from concurrent.futures import ThreadPoolExecutor
  from oracle import oracle
  n = len(phys)
  shards = [list(range(i, n, nthreads)) for i in range(min(nthreads, n))]

  def work(idx):
    oracle.rollout_legacy([phys[i] for i in idx], np.ascontiguousarray(acts[:, idx]), nsub)
  with ThreadPoolExecutor(len(shards)) as ex:
    list(ex.map(work, shards))
  return len(shards)


def cpu_baseline(model, nsub, q0, v0, w0, action_fn, nthreads, target_s):
  """Times the fp64 oracle (oracle/, test infrastructure) on the host cores over a bounded sample of the same
  workload (~target_s seconds of wall clock).  kind = 'port': MuJoCo itself is not installable here."""
  n = q0.shape[0]
  phys = make_oracles(model, q0, v0, w0)
  Tc = 2
  while True:                                   # calibration: double the chunk until it takes >= 0.3 s
    t0 = time.time()
    cores = threaded_rollout(phys, action_fn(0, Tc), nsub, nthreads)
    dtc = time.time() - t0
    if dtc >= 0.3 or Tc >= 256:
      break
    Tc *= 2
  rate = Tc * n / max(dtc, 1e-6)
  T = int(min(2000, max(3, target_s * rate / n)))
  acts = action_fn(2, T)
  t0 = time.time()
  threaded_rollout(phys, acts, nsub, nthreads)
  dt = time.time() - t0
  return dict(value=T * n / dt, unit='env-steps/s', cores=cores, kind='port',
              physics_steps_per_s=T * n * nsub / dt,
              sample='%d envs x %d env-steps x %d substeps (%.1f s wall), fp64 C restatement of mj_step (oracle/), one '
                     'thread per core' % (n, T, nsub, dt))


def mujoco_baseline(asset, nsub, q0, action_fn, nthreads, target_s):
  """cpu_baseline.kind = 'mujoco' (BASELINE.md 3.2): real MuJoCo through its Python bindings, one MjData per
  thread (the pattern of mujoco/thread_safety_test.py:56-73).  Only when `import mujoco` works."""
  import mujoco      # noqa: raises ImportError where the wheel is absent (this image)
  from concurrent.futures import ThreadPoolExecutor
  from dm_control_amd.suite import common
  m = mujoco.MjModel.from_xml_string(common.read_model(asset + '.xml'))
  n = q0.shape[0]
  datas = []
  for e in range(n):
    d = mujoco.MjData(m)
    d.qpos[:] = q0[e]
    mujoco.mj_forward(m, d)
    datas.append(d)
  shards = [list(range(i, n, nthreads)) for i in range(min(nthreads, n))]

  def run(acts):
    def work(idx):
      for t in range(acts.shape[0]):
        for i in idx:
          datas[i].ctrl[:] = acts[t, i]
          mujoco.mj_step(m, datas[i], nsub)
    with ThreadPoolExecutor(len(shards)) as ex:
      list(ex.map(work, shards))
  Tc = 2
  while True:
    t0 = time.time()
    run(action_fn(0, Tc))
    dtc = time.time() - t0
    if dtc >= 0.3 or Tc >= 256:
      break
    Tc *= 2
  rate = Tc * n / max(dtc, 1e-6)
  T = int(min(2000, max(3, target_s * rate / n)))
  acts = action_fn(2, T)
  t0 = time.time()
  run(acts)
  dt = time.time() - t0
  return dict(value=T * n / dt, unit='env-steps/s', cores=len(shards), kind='mujoco',
              sample='%d envs x %d env-steps x %d substeps (%.1f s wall), mujoco %s mj_step, one MjData per thread'
                     % (n, T, nsub, dt, mujoco.__version__))


def _committed_pmc(config):
  """Counters of the committed rocprofv3 PMC passes for this config (profiles/r02_pmc_cfg<N>.json), or None."""
  try:
    with open(os.path.join(ROOT, 'profiles', 'r02_pmc_cfg%d.json' % config)) as f:
      return json.load(f)
  except Exception:  # pylint: disable=broad-except
    return None


def rel_err(qg, qo):
  return np.abs(qg - qo).max(axis=1) / np.maximum(1.0, np.abs(qo).max(axis=1))


def parity(model, cfg, args, local_rank, q, v, w, acts, nthreads, only=None):
  """GPU vs the fp64 oracle on the first envs, same states and actions: open loop (both integrate on their own)
  and teacher-forced (the GPU state is overwritten by the oracle's before every env-step)."""
  from dm_control_amd.batch import BatchedPhysics
  from dm_control_amd.suite import common
  nsub = cfg['nsub']
  ne, T = q.shape[0], acts.shape[0]
  caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {}))
  caps.pop('precision', None)
  ops = make_oracles(model, q, v, w)
  res = {}
  # the fp64 instantiation of the same kernel, open loop: separates rounding (fp32 trajectories of a falling ragdoll
  # are chaotic) from logic (the fp64 kernel must stay on the oracle's trajectory)
  modes = [('open-loop', args.precision), ('teacher-forced', args.precision)] + ([('f64-open-loop', 64)] if args.precision == 32 else [])
  if nsub > 1:
    # forced before every PHYSICS step: the arithmetic error of one mj_step.  Between the forcings of 'teacher-forced' lie
    # n_sub_steps physics steps, over which a contact that fp32 and fp64 activate one step apart already changes the outcome
    modes.insert(2, ('teacher-forced-physics-step', args.precision))
  if only:
    modes = [m for m in modes if m[0] in only]
  for mode, prec in modes:
    chk, err = None, None
    # the fp64 scratch of the 62-dof models fits in LDS up to 32 .. 40 contacts: the parity leg lowers the cap if needed
    for kw in (caps, dict(caps, nconmax=32)):
      try:
        chk = BatchedPhysics(model, ne, device_id=local_rank, precision=prec, lanes_per_env=args.lanes if prec == args.precision else 0, **kw)
        break
      except Exception as ex:  # pylint: disable=broad-except
        err = repr(ex)[:200]
    if chk is None:
      res[mode] = dict(error=err)
      continue
    chk.set('qpos', q); chk.set('qvel', v); chk.set('qacc_warmstart', w)
    refs = [p.copy() for p in ops]
    worst = np.zeros(ne)
    every = mode == 'teacher-forced-physics-step'
    samples, edges = [], []
    for t in range(T):
      a = acts[t].astype(np.float64)
      chk.set_control(a)
      for k in range(nsub if every else 1):
        if (mode == 'teacher-forced' and t) or (every and (t or k)):
          chk.set('qpos', np.stack([p.qpos for p in refs]))
          chk.set('qvel', np.stack([p.qvel for p in refs]))
          # just-touching steps: the contact set the kernel finds AT THE FORCED STATE against the oracle's (margin 0: a
          # contact exists iff dist < 0, so a pair at |dist| below the working precision is a coin flip and MuJoCo's
          # dynamics are discontinuous across it).  Counted, and reported with and without them -- the GPU tests
          # (tests/test_gpu_suite.py, tests/test_gpu_parity.py) exclude exactly these steps.
          chk.forward()
          edge = chk.get('ncon')[:, 0].astype(np.int64) != np.array([p.ncon for p in refs])
          chk.set('qacc_warmstart', np.stack([p.qacc_warmstart for p in refs]))      # (mj_forward left its own solution there)
        else:
          edge = np.zeros(ne, bool)
        n = 1 if every else nsub
        chk.step(n)
        threaded_rollout(refs, a[None], n, nthreads)
        e = rel_err(chk.get('qpos'), np.stack([p.qpos for p in refs]))
        worst = np.maximum(worst, e)
        if mode != 'open-loop' and mode != 'f64-open-loop':
          samples.append(e)
          edges.append(edge)
    res[mode] = dict(max=float(worst.max()), median=float(np.median(worst)), p90=float(np.percentile(worst, 90)),
                     frac_le_1e4=float((worst <= 1e-4).mean()),
                     note='statistics over environments of the max over the run')
    if samples:      # teacher-forced modes: every comparison is an independent sample of the one-step error
      sm = np.concatenate(samples)
      res[mode]['per_step'] = dict(n=int(sm.size), median=float(np.median(sm)), p99=float(np.percentile(sm, 99)), max=float(sm.max()),
                                   frac_le_1e4=float((sm <= 1e-4).mean()))
      ed = np.concatenate(edges)
      keep = sm[~ed] if (~ed).any() else sm
      res[mode]['just_touching'] = dict(
          steps=int(ed.sum()), of=int(ed.size),
          max_excluding=float(keep.max()), frac_le_1e4_excluding=float((keep <= 1e-4).mean()),
          note='forced states at which the fp%d kernel and the fp64 oracle find different contact sets (a pair at |dist| '
               'below the working precision); `per_step` above INCLUDES them, the GPU parity tests exclude them' % prec)
    res[mode]['gpu_warnings'] = [int(x) for x in chk.get('warning').sum(axis=0)]
    res[mode]['oracle_warnings'] = [int(x) for x in np.sum([p.warning for p in refs], axis=0)]
    chk.close()
  return res


def quick_config_leg(cfgid, args, local_rank, nthreads, K=12, W=5, parity_envs=16, parity_steps=4, span=100):
  """A short leg of another BASELINE config for the default line's `extra`: the same workload, start states, actions and
  launch as `bench.py --config <cfgid>` (one Physics.step(n_sub_steps) launch per env-step, actions resident in HBM).  That
  run times the `span` = 100 launches after W = 5 warm-up steps, over which the ragdolls of configs 3 / 4 fall and a launch
  gets slower; this leg walks the same `span` launches and times K of them, evenly spread, each bracketed by its own pair
  of HIP events on the launch stream -- the same workload at a twelfth of the events.  Beside it: the nominal HBM roofline
  of the kernel and the fp32 error of ONE env-step from the oracle's state (teacher-forced, `parity_envs` environments x
  `parity_steps` steps)."""
  import torch
  from dm_control_amd.batch import BatchedPhysics, OUT
  from dm_control_amd.suite import common
  cfg = CONFIGS[cfgid]
  model = load_model(cfg['asset'])
  nsub, B = cfg['nsub'], cfg['batch']
  dev = torch.device('cuda', local_rank)
  caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {}))
  caps.pop('precision', None)
  t_leg = time.perf_counter()
  phys = BatchedPhysics(model, B, device_id=local_rank, precision=32, **caps)
  phys.set('qpos', initial_qpos(cfg, model, B, seed0=0, phys=phys))
  stream = torch.cuda.current_stream().cuda_stream
  rs = np.random.RandomState(1234)
  acts = torch.from_numpy(np.ascontiguousarray(rs.uniform(-1, 1, (W + span, B, model.nu)).astype(np.float32).transpose(0, 2, 1))).to(dev).contiguous()
  mask = 0
  for name in cfg['outputs']:
    mask |= OUT[name]
  phys.set_output_mask(mask)
  phys.forward()
  for t in range(W):
    phys.bind('ctrl', acts[t].data_ptr()); phys.step(nsub, stream=stream)
  torch.cuda.synchronize()
  q0, v0, w0 = phys.get('qpos'), phys.get('qvel'), phys.get('qacc_warmstart')
  timed = set(int(round(x)) for x in np.linspace(W, W + span - 1, K))
  pairs = []
  for t in range(W, W + span):
    if t in timed:
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
    phys.bind('ctrl', acts[t].data_ptr()); phys.step(nsub, stream=stream)
    if t in timed:
      e1.record(); pairs.append((e0, e1))
  torch.cuda.synchronize()
  K = len(pairs)
  kernel_ms = sum(a.elapsed_time(b) for a, b in pairs) / K
  elapsed = kernel_ms * 1e-3 * K      # (the timed launches only; the gaps between launches belong to the untimed ones too)
  warn = [int(x) for x in phys.get('warning').sum(axis=0)]
  info = phys.info()
  phys.close()
  achieved = cfg['algo_bytes'] * B / (kernel_ms * 1e-3) / 1e9
  out = dict(workload='BASELINE config %d: %s, batch %d, random actions, legacy Physics.step(%d), one launch per env-step'
                      % (cfgid, cfg['workload'], B, nsub),
             value=B * K / elapsed, unit='env-steps/s', physics_steps_per_s=B * K * nsub / elapsed, steps=K, warmup=W,
             launches_walked=span, ms_per_step=1e3 * elapsed / K, kernel_ms_avg=kernel_ms, dtype='f32', warnings_after_run=warn,
             roofline={'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                       'algorithmic_bytes_per_launch': cfg['algo_bytes'] * B, 'traffic': None,
                       'note': 'nominal bound; counters: bench.py --config %d' % cfgid},
             launch={k: info[k] for k in ('lanes_per_env', 'waves_per_block', 'envs_per_cu', 'static_id', 'work_queue')})
  try:
    ne = min(parity_envs, B)
    pa = np.random.RandomState(77).uniform(-1, 1, (parity_steps, ne, model.nu)).astype(np.float32)
    pargs = argparse.Namespace(precision=32, lanes=0)
    res = parity(model, cfg, pargs, local_rank, q0[:ne], v0[:ne], w0[:ne], pa, nthreads, only=('teacher-forced',))
    tf = res['teacher-forced']
    out['one_step_parity'] = dict(max=tf['per_step']['max'], median=tf['per_step']['median'], envs=ne, steps=parity_steps,
                                  just_touching=tf['just_touching']['steps'],
                                  note='fp32 kernel, one env-step (%d physics steps) from the fp64 oracle\'s state; rel. qpos error' % nsub)
  except Exception as ex:  # pylint: disable=broad-except
    out['one_step_parity'] = dict(error=repr(ex)[:300])
  out['leg_seconds'] = time.perf_counter() - t_leg
  return out


def env_step_leg(local_rank, B=4096, K=1000, W=20):
  """Row a1 (control.Environment.step, rl/control.py:99-127) on config 2's workload: the device-resident cheetah run
  environment -- action write, the physics launch, observation, reward, step counters and the per-environment restart -- in
  the timed region, next to the physics-only rate of the same batch object.  K = one episode (1000 steps): exactly one
  restart of every environment falls into the timed region.  Two task layers: suite/fused_env.py (`value`: the compiled
  one) and the hand-written suite/torch_env.py (`torch_env`: joint randomisation + the 200 settle steps of
  Cheetah.initialize_episode, suite/cheetah.py:63-76, at every restart)."""
  import torch
  from dm_control_amd.suite import torch_env
  t_leg = time.perf_counter()
  env = torch_env.make('cheetah', 'run', B, device_id=local_rank)
  dev = torch.device('cuda', local_rank)
  acts = torch.rand((W + K, B, env.model.nu), device=dev) * 2 - 1
  for t in range(W):
    env.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(W, W + K):
    obs, rew, done = env.step(acts[t])
  torch.cuda.synchronize()
  el = time.perf_counter() - t0
  stream = torch.cuda.current_stream().cuda_stream
  for t in range(W):      # physics only: the same action writes and launches, no task layer
    env.ctrl.copy_(acts[t].T); env.physics.step(env.n_sub_steps, stream=stream)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(W, W + K):
    env.ctrl.copy_(acts[t].T); env.physics.step(env.n_sub_steps, stream=stream)
  torch.cuda.synchronize()
  el_phys = time.perf_counter() - t0
  hand = dict(value=B * K / el, unit='env-steps/s', steps=K, ms_per_step=1e3 * el / K, physics_only_same_batch=B * K / el_phys,
              env_over_physics=el_phys / el, finite=bool(torch.isfinite(obs).all().item()),
              note='suite/torch_env.py: hand-written torch task layer, start states drawn on the device per episode (200 settle steps per restart)')
  env.close()
  # the same environment through suite/fused_env.py: the host port's task code compiled into the epilogue of the step kernel
  # (one launch per env-step), per-environment episode ends, restarts from a device-resident pool of start states drawn by
  # the port's own initialize_episode at construction
  from dm_control_amd.suite import fused_env
  fe = fused_env.make('cheetah', 'run', B, precision=32, device_id=local_rank, copy_outputs=False)
  for t in range(W):
    fe.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(W, W + K):
    fobs, frew, fdone = fe.step(acts[t])
  torch.cuda.synchronize()
  fel = time.perf_counter() - t0
  ctrl = fe._tensors['ctrl']      # pylint: disable=protected-access
  fe.restart(); fe.step(acts[0])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(W, W + K):
    ctrl.copy_(acts[t].T); fe.host_physics.batch.step(fe.n_sub_steps, stream=stream)
  torch.cuda.synchronize()
  fel_phys = time.perf_counter() - t0
  out = dict(workload="suite 'cheetah run' as a device-resident environment: obs (B, 17) + reward + done / first flags + per-environment restarts, "
                      "every step; %d steps = one episode of every environment" % K,
             value=B * K / fel, unit='env-steps/s', steps=K, warmup=W, ms_per_step=1e3 * fel / K, batch=B,
             physics_only_same_batch=B * K / fel_phys, env_over_physics=fel_phys / fel, launches_per_env_step=1 if fe.inline else 2,
             task_layer='suite/fused_env.py: get_observation / get_reward of suite/cheetah.py traced and generated as the epilogue of the step kernel'
                        if fe.inline else 'suite/fused_env.py: generated task kernel behind the physics launch',
             episodes_started=int(fe.episode.sum().item()), finite=bool(torch.isfinite(fobs).all().item()),
             torch_env=hand, leg_seconds=time.perf_counter() - t_leg)
  fe.close()
  return out


def main():
  args = parse()
  cfg = CONFIGS[args.config]
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    # bare `python bench.py --gpus N`: become the launcher of N ranks (one per GPU) and hand their exit code back
    cmd = launch_command(args, sys.argv[1:], free_port())
    print('bench.py: spawning %d ranks: %s' % (args.gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR='127.0.0.1')))
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  single_device = bool(os.environ.get('DMC_BENCH_SINGLE_DEVICE'))
  if os.environ.get('DMC_BENCH_DRYRUN'):      # harness test hook: the rank layout only, before anything touches a GPU
    check_world(args, world, args.gpus, single_device)
    # one write per rank: print() sends the text and the newline separately, and two ranks' lines can interleave
    sys.stdout.write(json.dumps({'dryrun': True, 'rank': rank, 'world': world, 'local_rank': local_rank, 'gpus': args.gpus}) + '\n')
    sys.stdout.flush()
    return
  import torch
  check_world(args, world, torch.cuda.device_count(), single_device)
  # DMC_BENCH_BACKEND=gloo + DMC_BENCH_SINGLE_DEVICE=1 exist only to exercise the N > 1
  # code path on a one-GPU box (all ranks share cuda:0); the driver uses nccl (= RCCL).
  backend = os.environ.get('DMC_BENCH_BACKEND', 'nccl')
  if single_device:
    local_rank = 0
  dist = None
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend, rank=rank, world_size=world)
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs a GPU: the batched step has no CPU fallback')
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  red_dev = dev if backend == 'nccl' else torch.device('cpu')
  from dm_control_amd.batch import BatchedPhysics, OUT
  from dm_control_amd.suite import common
  from dm_control_amd import sharding

  model = load_model(cfg['asset'])
  nsub = cfg['nsub']
  B = args.batch or cfg['batch']
  K = args.steps if args.steps is not None else (1000 if args.config == 2 else 100)
  W = args.warmup if args.warmup is not None else (50 if args.config == 2 else 5)
  tdtype = torch.float32 if args.precision == 32 else torch.float64
  caps = dict(common.DEFAULT_CAPS.get(cfg['asset'], {}))
  if caps.pop('precision', args.precision) != args.precision:
    raise SystemExit('%s runs in fp32 only' % cfg['asset'])
  shard = sharding.ShardedEnvBatch(B * world, dist, device=red_dev)      # contiguous env range of this rank
  assert shard.local_batch == B
  phys = BatchedPhysics(model, B, device_id=local_rank, precision=args.precision, lanes_per_env=args.lanes, **caps)
  q0 = initial_qpos(cfg, model, B, seed0=shard.lo, phys=phys)
  phys.set('qpos', q0)
  stream = torch.cuda.current_stream().cuda_stream
  # random actions, generated once, resident in HBM, SoA (T, nu, B)
  nact = W + K
  rs = np.random.RandomState(1234 + rank)
  actions_host = rs.uniform(-1, 1, (nact, B, model.nu)).astype(np.float32)
  actions = torch.from_numpy(np.ascontiguousarray(actions_host.transpose(0, 2, 1))).to(dev).to(tdtype).contiguous()
  mask = 0
  for name in cfg['outputs']:
    mask |= OUT[name]
  phys.set_output_mask(mask)
  if cfg['asset'] == 'cheetah':
    # settle 200 steps with zero control (Cheetah.initialize_episode), time = 0
    zero_ctrl = torch.zeros((model.nu, B), dtype=tdtype, device=dev)
    phys.bind('ctrl', zero_ctrl.data_ptr())
    phys.step(200, stream=stream)
    torch.cuda.synchronize()
    phys.set('time', np.zeros((B, 1)))
  else:
    phys.forward()
    torch.cuda.synchronize()
  q_start, v_start, w_start = phys.get('qpos'), phys.get('qvel'), phys.get('qacc_warmstart')

  def run(t0, n):
    for t in range(t0, t0 + n):
      phys.bind('ctrl', actions[t].data_ptr())
      phys.step(nsub, stream=stream)

  def barrier():
    if world > 1:
      dist.barrier()

  def max_over_ranks(x):
    if world == 1:
      return x
    t = torch.tensor([x], dtype=torch.float64, device=red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

  run(0, W)
  torch.cuda.synchronize()
  if cfg['asset'] != 'cheetah':
    # parity starts where the timed run starts (after the warm-up: the t = 0 drop of the start pose onto the floor,
    # |qacc| ~ 1e4, is a one-step transient that says nothing about the steady state)
    q_start, v_start, w_start = phys.get('qpos'), phys.get('qvel'), phys.get('qacc_warmstart')
  barrier()
  torch.cuda.synchronize()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t_start = time.perf_counter()
  ev0.record()
  run(W, K)
  ev1.record()
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  elapsed = max_over_ranks(time.perf_counter() - t_start)
  kernel_ms = ev0.elapsed_time(ev1) / K      # HIP events on the stream the kernel is launched on
  if args.pmc_child:                         # the parent reads the last K step_kernel dispatches from rocprofv3's csv
    phys.close()
    return
  q_end, v_end, w_end = phys.get('qpos'), phys.get('qvel'), phys.get('qacc_warmstart')
  warn = phys.get('warning').sum(axis=0)
  stats = dict(mean_ncon=float(phys.get('ncon').mean()), max_ncon=int(phys.get('ncon').max()),
               mean_nefc=float(phys.get('nefc').mean()), mean_solver_iter=float(phys.get('solver_iter').mean()))

  # ---- rollout leg: the same K env-steps, same resident action tensor, ONE launch per
  # chunk of <= 250 steps, per-step qpos/qvel/sensordata written to HBM (dmc_batch_rollout)
  chunk = min(250, K)
  nchunks = max(1, K // chunk)
  qs = torch.empty((chunk, model.nq, B), dtype=tdtype, device=dev)
  vs = torch.empty((chunk, model.nv, B), dtype=tdtype, device=dev)
  ss = torch.empty((chunk, max(1, model.nsensordata), B), dtype=tdtype, device=dev)
  phys.rollout(min(W, chunk) or 1, nsub, actions[0:].data_ptr(), qs.data_ptr(), vs.data_ptr(), ss.data_ptr(), stream=stream)
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  r0 = time.perf_counter()
  for c in range(nchunks):
    phys.rollout(chunk, nsub, actions[W + c*chunk:].data_ptr(), qs.data_ptr(), vs.data_ptr(), ss.data_ptr(), stream=stream)
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  rollout_elapsed = max_over_ranks(time.perf_counter() - r0)

  # ---- pipelined leg: the SAME B environments as P independent part-batches on P streams (part p's launch t+1 waits for
  # part p's launch t only, so the straggler tail of one part's queued launch overlaps the next part's launch).  A usage
  # mode of the existing API (P BatchedPhysics objects), reported beside `value`, never as `value`.
  pipe, pipes = None, []
  for P in [int(x) for x in str(args.pipeline).split(',') if x.strip()]:
    if not (P > 1 and B % P == 0):
      continue
    Bp = B // P
    parts, pstreams, pctrl = [], [], []
    for p in range(P):
      ph = BatchedPhysics(model, Bp, device_id=local_rank, precision=args.precision, lanes_per_env=args.lanes, **caps)
      ph.set('qpos', q0[p*Bp:(p+1)*Bp])
      ph.set_output_mask(mask)
      ps = torch.cuda.Stream()
      if cfg['asset'] == 'cheetah':
        zc = torch.zeros((model.nu, Bp), dtype=tdtype, device=dev)
        ph.bind('ctrl', zc.data_ptr())
        ph.step(200, stream=ps.cuda_stream)
        torch.cuda.synchronize()
        ph.set('time', np.zeros((Bp, 1)))
      else:
        ph.forward()
      parts.append(ph); pstreams.append(ps)
      pctrl.append(actions[:, :, p*Bp:(p+1)*Bp].contiguous())
    torch.cuda.synchronize()

    def run_parts(t0, n):
      for t in range(t0, t0 + n):
        for p in range(P):
          parts[p].bind('ctrl', pctrl[p][t].data_ptr())
          parts[p].step(nsub, stream=pstreams[p].cuda_stream)

    run_parts(0, W)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    p0 = time.perf_counter()
    run_parts(W, K)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    pipe_elapsed = max_over_ranks(time.perf_counter() - p0)
    pipe_q = np.concatenate([ph.get('qpos') for ph in parts], axis=0)
    pipe = dict(value=world * B * K / pipe_elapsed, unit='env-steps/s', parts=P, batch_per_part=Bp, steps=K,
                ms_per_env_step=1e3 * pipe_elapsed / K,
                max_abs_qpos_diff_vs_single_batch=float(np.max(np.abs(pipe_q - q_end))),
                note='same environments, actions and step count as `value`, stepped as %d independent part-batches on %d HIP '
                     'streams (one Physics.step() launch per part per env-step); the end state is compared with the '
                     'single-batch run' % (P, P))
    for ph in parts:
      ph.close()
    pipes.append(pipe)
  if pipes:      # `pipelined`: the best part count; every count measured is listed under `by_parts`
    pipe = dict(max(pipes, key=lambda d_: d_['value']), by_parts={str(d_['parts']): d_['value'] for d_ in pipes})

  # ---- agent-interface collectives (N > 1): actions scattered from rank 0, observations gathered to all ranks
  coll = None
  if world > 1:
    nobs = model.nq + model.nv + model.nsensordata
    glob_act = torch.rand((B * world, model.nu), device=red_dev) if rank == 0 else None
    obs = torch.cat([phys_t for phys_t in (torch.zeros((B, nobs), device=red_dev),)], dim=0)
    for _ in range(3):
      shard.scatter_actions(glob_act); shard.gather(obs)
    torch.cuda.synchronize(); barrier()
    c0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
      a_loc = shard.scatter_actions(glob_act)
      shard.gather(obs)
    torch.cuda.synchronize(); barrier()
    coll_ms = max_over_ranks((time.perf_counter() - c0) / reps * 1e3)
    # ---- the same exchange overlapped with the step: the shard as two part-batches, part p's all-gather / scatter in
    # flight while part p + 1's launch runs (sharding.PipelinedExchange); the "policy" on rank 0 consumes the gathered
    # observations of a part before it emits that part's next actions (a_{t+1} = pi(o_t))
    overlapped = None
    try:
      P2 = 2
      if B % P2 == 0:
        Bp = B // P2
        ex = sharding.PipelinedExchange(B * world, parts=P2, dist=dist, device=red_dev)
        pp, pbuf = [], []
        for p_ in range(P2):
          ph = BatchedPhysics(model, Bp, device_id=local_rank, precision=args.precision, lanes_per_env=args.lanes, **caps)
          tq = torch.zeros((model.nq, Bp), dtype=tdtype, device=dev); tv = torch.zeros((model.nv, Bp), dtype=tdtype, device=dev)
          tsn = torch.zeros((max(1, model.nsensordata), Bp), dtype=tdtype, device=dev); tc = torch.zeros((model.nu, Bp), dtype=tdtype, device=dev)
          ph.bind('qpos', tq.data_ptr()); ph.bind('qvel', tv.data_ptr()); ph.bind('sensordata', tsn.data_ptr()); ph.bind('ctrl', tc.data_ptr())
          ph.set('qpos', q_end[p_*Bp:(p_+1)*Bp]); ph.set('qvel', v_end[p_*Bp:(p_+1)*Bp])
          ph.set_output_mask(mask); ph.forward()
          pp.append(ph); pbuf.append((tq, tv, tsn, tc))
        torch.cuda.synchronize()
        table = torch.rand((8, world * Bp, model.nu), device=red_dev) * 2 - 1 if rank == 0 else None

        def observe(p_):
          tq, tv, tsn, _ = pbuf[p_]
          return torch.cat([tq, tv, tsn[:model.nsensordata]], dim=0).T.to(torch.float32).to(red_dev)

        def policy(t, o):
          return table[t % 8] + 0.0 * o[:, :1]      # (depends on the gathered observations: cannot be issued before them)

        def loop(n, t0_):
          for t in range(t0_, t0_ + n):
            for p_ in range(P2):
              a = ex.scatter_wait(p_)
              pbuf[p_][3].copy_(a.T.to(dev).to(tdtype))
              pp[p_].step(nsub, stream=stream)
              ex.gather_async(p_, observe(p_))
            for p_ in range(P2):
              o = ex.gather_wait(p_)
              ex.scatter_async(p_, policy(t, o) if rank == 0 else None, model.nu)
        for p_ in range(P2):
          ex.scatter_async(p_, table[0] if rank == 0 else None, model.nu)
        Ko = min(K, 50)
        loop(3, 0)
        torch.cuda.synchronize(); barrier()
        o0 = time.perf_counter()
        loop(Ko, 3)
        torch.cuda.synchronize(); barrier()
        ov_s = max_over_ranks(time.perf_counter() - o0)
        for p_ in range(P2):
          ex.scatter_wait(p_)
        overlapped = dict(value=world * B * Ko / ov_s, unit='env-steps/s', steps=Ko, parts=P2, ms_per_env_step=1e3 * ov_s / Ko,
                          note='policy on rank 0; the shard of every rank as %d part-batches, the exchange of one part in flight '
                               'while the other part steps (sharding.PipelinedExchange)' % P2)
        for ph in pp:
          ph.close()
    except Exception as ex_:  # pylint: disable=broad-except
      overlapped = dict(error=repr(ex_)[:300])
    coll = dict(ms_per_env_step=coll_ms, backend=backend, n_ranks_seen=dist.get_world_size(),
                env_steps_per_s_with_exchange_overlapped=(overlapped or {}).get('value'), overlapped=overlapped,
                action_bytes=int(B * world * model.nu * 4), observation_bytes=int(B * world * nobs * 4),
                env_steps_per_s_with_exchange=world * B / (elapsed / K + coll_ms * 1e-3),
                note='scatter of (B, nu) fp32 actions from rank 0 + all_gather of (B, nq+nv+nsensordata) fp32 '
                     'observations, serialised with the step (no overlap): the agent-interface cost when the policy '
                     'lives on one rank; `value` is the no-collective number (policy co-located with each shard)')

  if rank == 0:
    value = world * B * K / elapsed
    algo = cfg['algo_bytes']
    achieved = algo * B / (kernel_ms * 1e-3) / 1e9
    live, why = (None, 'N > 1') if world > 1 else collect_pmc(args, 30)
    pmc_source = 'live: rocprofv3 --pmc passes of this run (bench.py --pmc-child, 30 launches per pass)'
    if live and 'FETCH_SIZE' in live and 'WRITE_SIZE' in live:
      pmc = dict(live)
      # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM): x 2 on reads
      pmc['hbm_bytes_per_launch'] = dict(fetch_raw=live['FETCH_SIZE'] * 1024, fetch_corrected=2 * live['FETCH_SIZE'] * 1024,
                                         write=live['WRITE_SIZE'] * 1024,
                                         total_corrected=(2 * live['FETCH_SIZE'] + live['WRITE_SIZE']) * 1024)
      if live.get('GRBM_GUI_ACTIVE') and live.get('kernel_us_under_pmc:SQ_INSTS_VALU'):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs and counts every cycle the graphics pipe is busy -- the dispatch
        # and the end-of-kernel drain around the kernel too -- so busy cycles / kernel time over-estimates the clock
        # (round 3 printed 2.51 GHz, above the chip's maximum): the estimate is capped at the 2.4 GHz peak engine clock
        # of MI355X_MICROARCH.md and the raw figure kept beside it.
        raw = live['GRBM_GUI_ACTIVE'] / 8 / (live['kernel_us_under_pmc:SQ_INSTS_VALU'] * 1e3)
        pmc['clock_ghz_raw'] = raw
        pmc['clock_ghz'] = min(raw, MAX_CLOCK_GHZ)
      pmc['kernel_us'] = live.get('kernel_us_under_pmc:SQ_INSTS_VALU')
    else:
      pmc = _committed_pmc(args.config)
      pmc_source = 'committed: profiles/r02_pmc_cfg%d.json (not measured in this run: %s)' % (args.config, why)
    roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            # HBM bytes per launch: separate --pmc FETCH_SIZE / WRITE_SIZE passes, reads x 2 as MI355X_MICROARCH.md prescribes
            'traffic': (pmc.get('hbm_bytes_per_launch') or {}).get('total_corrected') if pmc else None,
            'traffic_source': pmc_source if pmc else None,
            'traffic_detail': pmc.get('hbm_bytes_per_launch') if pmc else None,
            'traffic_over_algorithmic': ((pmc.get('hbm_bytes_per_launch') or {}).get('total_corrected', 0) / (algo * B)) if pmc else None,
            'kernel_ms_avg': kernel_ms, 'algorithmic_bytes_per_launch': algo * B,
            'note': 'nominal bound (SURVEY 8(d)): compulsory HBM traffic is %d B per env per launch; the kernel is '
                    'VALU-issue / latency bound, see roofline_issue' % algo}
    out = {
        'metric': METRIC,
        'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
        'ms_per_step': 1e3 * elapsed / K, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32' if args.precision == 32 else 'f64', 'data': 'synthetic',
        'config': {'workload': "BASELINE config %d: %s, batch %d per GPU, random actions U(-1,1)^%d, legacy "
                               "Physics.step(%d), one fused kernel launch per env-step" % (args.config, cfg['workload'], B, model.nu, nsub),
                   'baseline_config': args.config, 'batch_per_gpu': B, 'global_batch': B * world, 'n_sub_steps': nsub,
                   'sharding': 'contiguous env ranges, ranks %d' % world, 'info': phys.info()},
        'physics_steps_per_s': value * nsub,
        'roofline': roof,
        'workload_stats': stats,
        'warnings_after_run': [int(x) for x in warn],
        'rollout': {'value': world * B * nchunks * chunk / rollout_elapsed, 'unit': 'env-steps/s',
                    'steps': nchunks * chunk, 'launches': nchunks,
                    'note': 'same workload via dmc_batch_rollout: per-step actions read from / per-step '
                            'qpos,qvel,sensordata written to HBM, no launch per step; `value` above is the '
                            'host-in-the-loop mode (one Physics.step() launch per env-step)'},
    }
    if pmc and pmc.get('SQ_INSTS_VALU'):
      # what actually limits the kernel: VALU issue.  achieved = wave64 VALU instructions per launch (counter) x the issue
      # cost of one instruction / (1024 SIMDs x clock x kernel time of THIS run's HIP events: counter collection slows
      # the launch down, the instruction count does not depend on it).  Primary convention: 2 cycles per fp32 wave64
      # instruction (MI355X_MICROARCH.md); `frac_quad_cycle` is round 2's convention (one SQ quad-cycle = 4 cycles).
      clk = pmc.get('clock_ghz') or 2.4
      slots = SIMDS * clk * 1e9 * kernel_ms * 1e-3
      issue = pmc['SQ_INSTS_VALU'] * VALU_CYCLES_PER_INST / slots
      out['roofline_issue'] = {'bound': 'valu_issue', 'achieved': issue, 'peak': 1.0, 'unit': 'fraction of VALU issue cycles',
                               'frac': issue, 'cycles_per_wave64_valu_inst': VALU_CYCLES_PER_INST,
                               'frac_quad_cycle': pmc['SQ_INSTS_VALU'] * 4 / slots,
                               'valu_insts_per_launch': pmc['SQ_INSTS_VALU'], 'clock_ghz': clk, 'clock_ghz_raw': pmc.get('clock_ghz_raw'),
                               'clock_capped_at_peak': bool(pmc.get('clock_ghz_raw') and pmc['clock_ghz_raw'] > MAX_CLOCK_GHZ),
                               'kernel_us_under_pmc': pmc.get('kernel_us'), 'kernel_us_live': kernel_ms * 1e3,
                               'wait_any_over_wave_cycles': (pmc['SQ_WAIT_ANY'] / pmc['SQ_WAVE_CYCLES'])
                               if pmc.get('SQ_WAIT_ANY') and pmc.get('SQ_WAVE_CYCLES') else None,
                               'source': pmc_source}
    if pipe:
      out['pipelined'] = pipe
    if coll:
      out['collectives'] = coll
    nthreads = os.cpu_count() or 1
    # ---- parity vs the CPU oracle on the first envs (checker only, untimed) -------
    try:
      ne = min(args.parity_envs if args.parity_envs is not None else cfg['parity_envs'], B)
      T = args.parity_steps if args.parity_steps is not None else cfg['parity_steps']
      if T > 0 and ne > 0:
        pa = np.random.RandomState(77).uniform(-1, 1, (T, ne, model.nu)).astype(np.float32)
        res = parity(model, cfg, args, local_rank, q_start[:ne], v_start[:ne], w_start[:ne], pa, nthreads)
        out['max_rel_qpos_err_vs_cpu'] = res['open-loop']['max']
        out['parity'] = dict(res, envs=ne, steps=T, n_sub_steps=nsub,
                             oracle='fp64 C restatement (oracle/); kinematics pinned on reference-held MuJoCo output (tests/golden/cmu2019_mocap.json), '
                                    'dynamics on the reference KATs + independent solver-optimum pins; trajectories unpinned vs real MuJoCo')
    except Exception as ex:  # pylint: disable=broad-except
      out['max_rel_qpos_err_vs_cpu'] = None
      out['parity_error'] = repr(ex)
    # ---- CPU baseline (rank 0, N = 1 only; bounded sample) ----------------------------
    if world == 1 and not args.no_cpu_baseline:
      nb = min(args.cpu_envs, B)

      def action_fn(t0, n):
        return np.random.RandomState(99 + t0).uniform(-1, 1, (n, nb, model.nu))
      try:
        try:
          out['cpu_baseline'] = mujoco_baseline(cfg['asset'], nsub, q_end[:nb], action_fn, nthreads, args.cpu_seconds)
        except ImportError:
          out['cpu_baseline'] = cpu_baseline(model, nsub, q_end[:nb], v_end[:nb], w_end[:nb], action_fn, nthreads, args.cpu_seconds)
      except Exception as ex:  # pylint: disable=broad-except
        out['cpu_baseline'] = {'value': None, 'error': repr(ex)}
    do_extra = args.extra if args.extra is not None else int(args.config == 2 and world == 1 and args.batch is None and args.precision == 32)
    if do_extra:
      # the other BASELINE configs and row a1 in the driver's line; each leg reports its own seconds
      extra = {'configs': {}}
      for c in (3, 4, 5):
        try:
          extra['configs'][str(c)] = quick_config_leg(c, args, local_rank, nthreads, K=max(10, args.extra_steps))
        except Exception as ex:  # pylint: disable=broad-except
          extra['configs'][str(c)] = {'error': repr(ex)[:300]}
      try:
        extra['env_step'] = env_step_leg(local_rank)
      except Exception as ex:  # pylint: disable=broad-except
        extra['env_step'] = {'error': repr(ex)[:300]}
      out['extra'] = extra
    print(json.dumps(out), flush=True)
  phys.close()
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
