#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched physics step (BASELINE.json metric).

Workload (BASELINE.json configs[1], SURVEY.md 8(d) cfg 2): suite 'cheetah run',
batch 4096 per GPU, fp32, random actions U(-1,1)^6 generated once and resident
in HBM.  One "step" = one `Physics.step()` over the whole batch = ONE launch of
the fused HIP kernel (n_sub_steps = 1 for cheetah, suite/cheetah.py:48), with
legacy_step semantics (engine.py:147-162).  Initial states follow
Cheetah.initialize_episode (suite/cheetah.py:63-76): limited joints ~ U(range),
200 settle steps with zero control, time = 0 (untimed).

N > 1: one process per GPU (torch.distributed / RCCL only for the barrier and
the max-over-ranks timing; environments are independent, so there is no
data-path collective) -- weak scaling, 4096 envs per GPU.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 260          # cheetah fp32 SoA, state + ctrl in, state + sensordata out
PMC_TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec
BATCH_PER_GPU = 4096


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=1000)
  ap.add_argument('--warmup', type=int, default=50)
  ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='envs per GPU')
  ap.add_argument('--lanes', type=int, default=int(os.environ.get('DMC_LANES', '0')))
  ap.add_argument('--precision', type=int, default=32)
  ap.add_argument('--parity-envs', type=int, default=32)
  ap.add_argument('--parity-steps', type=int, default=200)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-envs', type=int, default=4096)
  return ap.parse_args()


def cheetah_model():
  from dm_control_amd import mjcf_compiler
  with open(os.path.join(ROOT, 'dm_control_amd', 'suite', 'assets', 'cheetah.xml')) as f:
    return mjcf_compiler.compile_xml(f.read())


def initial_qpos(model, n, seed0):
  """Cheetah.initialize_episode: qpos[is_limited] ~ U(lower, upper), per-env seed."""
  q = np.tile(model.qpos0, (n, 1))
  lim = model.jnt_limited == 1
  lo, hi = model.jnt_range[lim].T
  for e in range(n):
    q[e, lim] = np.random.RandomState(seed0 + e).uniform(lo, hi)
  return q


def cpu_baseline(model, q0, action_fn, nthreads, target_s=8.0, max_steps=1000):
  """Times the fp64 oracle (oracle/, test infrastructure) on host cores over a
  bounded sample of the same workload (~target_s seconds of wall clock on all
  host cores).  kind = 'port': MuJoCo itself is not installable here."""
  from concurrent.futures import ThreadPoolExecutor
  from oracle import oracle
  B = q0.shape[0]
  phys = []
  for e in range(B):
    p = oracle.OraclePhysics(model)
    p.qpos[:] = q0[e]
    p.forward()
    phys.append(p)
  shards = [list(range(i, B, nthreads)) for i in range(nthreads)]
  shards = [s for s in shards if s]

  def run(acts):
    def work(idx):
      oracle.rollout_legacy([phys[i] for i in idx], np.ascontiguousarray(acts[:, idx]))
    with ThreadPoolExecutor(len(shards)) as ex:
      list(ex.map(work, shards))
  run(np.zeros((200, B, model.nu)))   # settle (untimed)
  t0 = time.time()
  run(action_fn(0, 20))               # calibration chunk
  rate = 20 * B / max(time.time() - t0, 1e-6)
  T = int(min(max_steps, max(50, target_s * rate / B)))
  acts = action_fn(20, T)
  t0 = time.time()
  run(acts)
  dt = time.time() - t0
  return dict(value=T * B / dt, unit='env-steps/s', cores=len(shards), kind='port',
              sample='%d envs x %d steps (%.1f s wall), fp64 C restatement of mj_step (oracle/), one thread per core'
                     % (B, T, dt))


def _pmc_traffic():
  """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None."""
  try:
    with open(PMC_TRAFFIC_FILE) as f:
      return json.load(f)['hbm_bytes_per_launch']
  except Exception:  # pylint: disable=broad-except
    return None


def main():
  args = parse()
  import torch
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  # DMC_BENCH_BACKEND=gloo + DMC_BENCH_SINGLE_DEVICE=1 exist only to exercise the N > 1
  # code path on a one-GPU box (all ranks share cuda:0); the driver uses nccl (= RCCL).
  backend = os.environ.get('DMC_BENCH_BACKEND', 'nccl')
  if os.environ.get('DMC_BENCH_SINGLE_DEVICE'):
    local_rank = 0
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

  model = cheetah_model()
  B, K, W = args.batch, args.steps, args.warmup
  tdtype = torch.float32 if args.precision == 32 else torch.float64
  phys = BatchedPhysics(model, B, device_id=local_rank, precision=args.precision, lanes_per_env=args.lanes)
  q0 = initial_qpos(model, B, seed0=rank * B)
  phys.set('qpos', q0)
  stream = torch.cuda.current_stream().cuda_stream
  # random actions, generated once, resident in HBM, SoA (T, nu, B)
  nact = W + K
  rs = np.random.RandomState(1234 + rank)
  actions_host = rs.uniform(-1, 1, (nact, B, model.nu)).astype(np.float32)
  actions = torch.from_numpy(np.ascontiguousarray(actions_host.transpose(0, 2, 1))).to(dev).to(tdtype).contiguous()
  zero_ctrl = torch.zeros((model.nu, B), dtype=tdtype, device=dev)
  # outputs the cheetah task reads: qpos, qvel (state) + sensordata (speed)
  phys.set_output_mask(OUT['sensor'])
  # settle 200 steps with zero control (Cheetah.initialize_episode), time = 0
  phys.bind('ctrl', zero_ctrl.data_ptr())
  phys.step(200, stream=stream)
  torch.cuda.synchronize()
  phys.set('time', np.zeros((B, 1)))
  q_settled = phys.get('qpos')
  v_settled = phys.get('qvel')
  w_settled = phys.get('qacc_warmstart')

  def run(t0, n):
    for t in range(t0, t0 + n):
      phys.bind('ctrl', actions[t].data_ptr())
      phys.step(1, stream=stream)

  def barrier():
    if world > 1:
      torch.distributed.barrier()

  run(0, W)
  torch.cuda.synchronize()
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
  elapsed = time.perf_counter() - t_start
  kernel_ms = ev0.elapsed_time(ev1) / K
  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
  # ---- rollout leg: the same K env-steps, same resident action tensor, ONE launch per
  # chunk of <= 250 steps, per-step qpos/qvel/sensordata written to HBM (dmc_batch_rollout)
  chunk = min(250, K)
  nchunks = max(1, K // chunk)
  qs = torch.empty((chunk, model.nq, B), dtype=tdtype, device=dev)
  vs = torch.empty((chunk, model.nv, B), dtype=tdtype, device=dev)
  ss = torch.empty((chunk, model.nsensordata, B), dtype=tdtype, device=dev)
  phys.rollout(min(W, chunk) or 1, 1, actions[0:].data_ptr(), qs.data_ptr(), vs.data_ptr(), ss.data_ptr(), stream=stream)
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  r0 = time.perf_counter()
  for c in range(nchunks):
    phys.rollout(chunk, 1, actions[W + c*chunk:].data_ptr(), qs.data_ptr(), vs.data_ptr(), ss.data_ptr(), stream=stream)
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  rollout_elapsed = time.perf_counter() - r0
  if world > 1:
    t = torch.tensor([rollout_elapsed], dtype=torch.float64, device=red_dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    rollout_elapsed = float(t.item())
  warn = phys.get('warning').sum(axis=0)

  if rank == 0:
    value = world * B * K / elapsed
    achieved = ALGO_BYTES_PER_STEP * B / (kernel_ms * 1e-3) / 1e9
    out = {
        'metric': 'env-steps/s (whole node) at batch 4096; max rel qpos error vs CPU',
        'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
        'ms_per_step': 1e3 * elapsed / K, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32' if args.precision == 32 else 'f64', 'data': 'synthetic',
        'config': {'workload': "suite 'cheetah run', batch %d per GPU, random actions U(-1,1)^6, "
                               "legacy Physics.step(1), one fused kernel launch per step" % B,
                   'batch_per_gpu': B, 'n_sub_steps': 1, 'info': phys.info()},
        'physics_steps_per_s': value,
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': _pmc_traffic(),
                     'kernel_ms_avg': kernel_ms, 'algorithmic_bytes_per_launch': ALGO_BYTES_PER_STEP * B,
                     'note': 'latency/LDS/VALU-bound kernel: compulsory HBM traffic is 260 B per env-step'},
        'warnings_after_run': [int(x) for x in warn],
        'rollout': {'value': world * B * nchunks * chunk / rollout_elapsed, 'unit': 'env-steps/s',
                    'steps': nchunks * chunk, 'launches': nchunks,
                    'note': 'same workload via dmc_batch_rollout: per-step actions read from / per-step '
                            'qpos,qvel,sensordata written to HBM, no launch per step; `value` above is the '
                            'host-in-the-loop mode (one Physics.step() launch per env-step)'},
    }
    # ---- parity vs the CPU oracle on the first envs (checker only, untimed) -------
    try:
      ne, T = min(args.parity_envs, B), min(args.parity_steps, K)
      from oracle import oracle
      ops = []
      for e in range(ne):
        p = oracle.OraclePhysics(model)
        p.qpos[:] = q_settled[e]; p.qvel[:] = v_settled[e]; p.qacc_warmstart[:] = w_settled[e]
        p.step1()
        ops.append(p)
      chk = BatchedPhysics(model, ne, device_id=local_rank, precision=args.precision, lanes_per_env=args.lanes)
      chk.set('qpos', q_settled[:ne]); chk.set('qvel', v_settled[:ne]); chk.set('qacc_warmstart', w_settled[:ne])
      err = 0.0
      for t in range(T):
        a = actions_host[t, :ne].astype(np.float64)
        chk.set_control(a)
        chk.step()
        oracle.rollout_legacy(ops, a[None])
        qg = chk.get('qpos')
        qo = np.stack([p.qpos for p in ops])
        err = max(err, float((np.abs(qg - qo).max(axis=1) / np.maximum(1.0, np.abs(qo).max(axis=1))).max()))
      out['max_rel_qpos_err_vs_cpu'] = err
      out['parity'] = {'envs': ne, 'steps': T, 'mode': 'open-loop', 'oracle': 'fp64 C restatement (parity unpinned vs real MuJoCo)'}
      chk.close()
    except Exception as ex:  # pylint: disable=broad-except
      out['max_rel_qpos_err_vs_cpu'] = None
      out['parity_error'] = repr(ex)
    # ---- CPU baseline (rank 0, N = 1 only; bounded sample) ----------------------------
    if world == 1 and not args.no_cpu_baseline:
      try:
        nthreads = os.cpu_count() or 1
        nb = min(args.cpu_envs, B)

        def action_fn(t0, n):
          return np.random.RandomState(99 + t0).uniform(-1, 1, (n, nb, model.nu))
        cb = cpu_baseline(model, q0[:nb], action_fn, nthreads)
        out['cpu_baseline'] = cb
      except Exception as ex:  # pylint: disable=broad-except
        out['cpu_baseline'] = {'value': None, 'error': repr(ex)}
    print(json.dumps(out), flush=True)
  phys.close()
  if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
