import os

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets')


# Contact / constraint-row caps per domain where the library default (16 contacts) is too tight;
# the humanoid's are free: its LDS footprint puts 4 environments on a CU either way.  The
# model-specialised kernels are baked for exactly these caps (dm_control_amd/build.py).
# humanoid_CMU (nv = 62): one environment fills a CU's LDS in fp32 (98 KiB of scratch + 58 KiB of
# tables at 32 contacts); the fp64 scratch does not fit, so the domain runs the fp32 kernel.
DEFAULT_CAPS = {'humanoid': dict(nconmax=24), 'humanoid_CMU': dict(nconmax=32, precision=32),
                'cmu_2019_position_floor': dict(nconmax=32, precision=32),
                'soccer_2v2_boxhead': dict(nconmax=24),   # BASELINE config 5 physics (assets/)
                'stacker': dict(nconmax=48)}   # four boxes in a heap + a folded arm: many simultaneous contacts   # BASELINE config 4 physics (assets/)


def physics_kwargs(domain, user_kwargs):
  kw = dict(DEFAULT_CAPS.get(domain, {}))
  kw.update(user_kwargs or {})
  return kw


def read_model(filename):
  """Physics-only restatement of the reference model of the same name."""
  with open(os.path.join(_ASSETS, filename)) as f:
    return f.read()
