import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


# The test tier builds hundreds of small throw-away models; the product default (specialise.py: compile every unseen
# model's kernel in a background thread) would queue a hipcc run for each.  Tests run the generic kernel unless they ask
# for a specialised one (tests/test_specialise.py covers the default mode explicitly).
os.environ.setdefault('DMC_SPECIALISE', 'cached')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
  """`gpu` tests need a HIP device and the built library: skip (not fail) where either is missing, so that a plain
  `pytest tests` works on a CPU-only box."""
  lib = os.path.join(ROOT, 'dm_control_amd', 'libdmc_hip.so')
  try:
    import torch
    have = torch.cuda.is_available() and os.path.exists(lib)
  except Exception:  # pylint: disable=broad-except
    have = False
  if have:
    return
  skip = pytest.mark.skip(reason='needs an MI355X and dm_control_amd/libdmc_hip.so')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


@pytest.fixture(scope='session')
def oracle_lib():
  from oracle import oracle
  return oracle.lib()


@pytest.fixture
def oracle_backend(monkeypatch):
  """Runs the host stack (Physics facade, Environment, suite tasks) on the CPU oracle for one test:
  `dm_control_amd.physics.BatchedPhysics` is replaced by tests/oracle_backend.OracleBatch.  Test
  infrastructure only -- the product has no CPU path."""
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import oracle_backend as ob
  from dm_control_amd import physics
  monkeypatch.setattr(physics, 'BatchedPhysics', ob.OracleBatch)
  return ob
