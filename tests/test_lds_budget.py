"""Host-only guard of the LDS budget: the residency per CU that DESIGN.md section 3 claims for the BASELINE models follows
from the scratch layout (dm_control_amd/csrc/step_layout.h) and the caps the suite ships; a field added to the wrong
list silently costs a resident environment (measured: -11 % when the 27-dof humanoid went from 8 to 6 per CU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import lds_report  # noqa: E402

LDS_PER_CU = 160 * 1024
OPTS_HEADER = 256      # StepOpts in LDS, rounded up


def _fit(r, waves, envs_per_wave=1, elem=4):
  tables = OPTS_HEADER + r['n_mi'] * 4 + r['n_mr_lds'] * elem
  env = r['n_sr'] * elem + r['n_si'] * 4
  block = tables + waves * envs_per_wave * env
  return LDS_PER_CU // block, env, tables


@pytest.mark.parametrize('spec,waves,blocks,global_kb', [
    ('cmu_2019_position_floor:48', 5, 1, 30),      # BASELINE config 4: FIVE environments per CU in one 5-wave workgroup (offload level 3; +8 %)
    ('humanoid_CMU:96', 4, 1, 20),                 # suite humanoid_CMU at its production contact cap
    ('humanoid:24', 4, 2, 4),                      # BASELINE config 3: 2 workgroups x 4 = 8 per CU
    ('soccer_2v2_boxhead:24', 1, 4, 4),            # BASELINE config 5 (the reference's composed model: 113 sensors, 62 geoms; 5 with the round-2 restatement)
])
def test_fp32_residency_of_the_baseline_models(spec, waves, blocks, global_kb):
  r = lds_report.report(spec)
  fit, env, tables = _fit(r, waves)
  assert fit >= blocks, (spec, 'env bytes', env, 'table bytes', tables, 'workgroups per CU', fit)
  # what left LDS is in the per-environment global scratch, in 128-byte granules
  assert r['n_gs'] * 4 >= global_kb * 1024 and r['n_gs'] % 32 == 0, r['n_gs']


def test_small_models_keep_everything_in_lds_and_fp64_fits_for_the_62_dof_models():
  assert lds_report.report('cheetah')['n_gs'] == 0
  for spec in ('cmu_2019_position_floor:48', 'humanoid_CMU:96'):
    r = lds_report.report(spec)
    fit, env, tables = _fit(r, 2, elem=8)
    assert fit >= 1, (spec, env, tables)      # fp64 parity runs of these models: one 2-wave workgroup = two environments per CU
