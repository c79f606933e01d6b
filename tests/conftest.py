import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


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
