"""bench.py's multi-GPU harness on CPU: `--gpus N` spawns N ranks itself, and a world that is not N ranks fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_is_one_rank_per_gpu_on_localhost():
  args = bench.parse(['--gpus', '4', '--config', '4'])
  cmd = bench.launch_command(args, ['--gpus', '4', '--config', '4'], 12345)
  assert cmd[1:3] == ['-m', 'torch.distributed.run']
  assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and '--nnodes=1' in cmd
  assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '12345'
  assert cmd[-5].endswith('bench.py') and cmd[-4:] == ['--gpus', '4', '--config', '4']


def test_world_must_match_gpus():
  args = bench.parse(['--gpus', '2'])
  bench.check_world(args, 2, 2, False)
  bench.check_world(args, 2, 1, True)                  # single-device harness mode
  with pytest.raises(SystemExit, match='WORLD_SIZE=1'):
    bench.check_world(args, 1, 8, False)
  with pytest.raises(SystemExit, match='only 1 GPU'):
    bench.check_world(args, 2, 1, False)


def _run(argv, **env):
  e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
  e.update(env, DMC_BENCH_DRYRUN='1')
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=e, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=300)


def test_bare_gpus_2_spawns_two_ranks():
  r = _run(['--gpus', '2', '--config', '4'])
  assert r.returncode == 0, r.stderr.decode()[-400:]
  rows = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith('{')]
  assert sorted((x['rank'], x['local_rank'], x['world']) for x in rows) == [(0, 0, 2), (1, 1, 2)]


def test_launcher_world_mismatch_fails_loudly():
  r = _run(['--gpus', '2'], WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
  assert r.returncode != 0 and b'--gpus 2 but the launcher started WORLD_SIZE=1' in r.stderr


def test_default_is_one_gpu_no_spawn():
  r = _run([])
  assert r.returncode == 0
  rows = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith('{')]
  assert len(rows) == 1 and rows[0]['world'] == 1 and rows[0]['gpus'] == 1
