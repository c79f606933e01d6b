import os

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets')


def read_model(filename):
  """Physics-only restatement of the reference model of the same name."""
  with open(os.path.join(_ASSETS, filename)) as f:
    return f.read()
