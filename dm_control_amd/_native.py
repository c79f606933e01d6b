"""ctypes binding of libdmc_hip.so (include/dmc_batch.h).  There is no CPU
fallback: if the HIP library is missing or fails to load, importing raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_VARIANT = 'prof' if os.environ.get('DMC_USE_PROF') else os.environ.get('DMC_LIB_VARIANT')
LIB_PATH = os.path.join(_HERE, 'libdmc_hip_%s.so' % _VARIANT if _VARIANT else 'libdmc_hip.so')

EXPORTS = [
    'dmc_last_error', 'dmc_model_create', 'dmc_model_destroy', 'dmc_batch_create', 'dmc_batch_create_caps',
    'dmc_batch_destroy', 'dmc_batch_step', 'dmc_batch_rollout', 'dmc_batch_forward', 'dmc_batch_reset',
    'dmc_batch_field_rows', 'dmc_batch_get', 'dmc_batch_set', 'dmc_batch_get_int',
    'dmc_batch_set_int', 'dmc_batch_device_ptr', 'dmc_batch_bind',
    'dmc_batch_set_output_mask', 'dmc_batch_set_opt_int', 'dmc_batch_set_opt_real',
    'dmc_batch_set_model_real',
    'dmc_batch_step1', 'dmc_batch_step2', 'dmc_batch_sync', 'dmc_batch_invalidate', 'dmc_batch_invalidate_async', 'dmc_batch_info', 'dmc_batch_time_steps',
    'dmc_batch_enable_profiling', 'dmc_batch_get_timer', 'dmc_batch_set_step_probe', 'dmc_batch_set_async', 'dmc_batch_get_async', 'dmc_batch_get_wait', 'dmc_batch_get_staged',
    'dmc_batch_debug_enable', 'dmc_batch_debug_get', 'dmc_batch_prof_enable',
    'dmc_batch_prof_get', 'dmc_gather_create', 'dmc_gather_destroy', 'dmc_gather_run',
    'dmc_batch_set_env_geoms', 'dmc_env_geom_pack', 'dmc_batch_wave_trace', 'dmc_batch_randomize_joints',
    'dmc_batch_attach_specialised', 'dmc_batch_set_task_args', 'dmc_batch_enable_task',
]

_lib = None


class NativeError(RuntimeError):
  pass


def lib():
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise NativeError(
        'libdmc_hip.so not built: run `python -m dm_control_amd.build` '
        '(or __graft_entry__.build()).  There is no CPU fallback.')
  # PyTorch-ROCm bundles its own HIP runtime; load it first so that this library
  # and torch share ONE libamdhip64 (two runtimes in a process cannot both see
  # the GPU).  torch is only plumbing here (device memory, streams, RCCL).
  try:
    import torch  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
  except ImportError:
    pass
  L = ctypes.CDLL(LIB_PATH)
  vp, ci, cd, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char_p
  L.dmc_last_error.restype = cs
  L.dmc_model_create.argtypes = [vp, ci, vp, ci, ctypes.POINTER(vp)]
  L.dmc_model_destroy.argtypes = [vp]
  L.dmc_batch_create.argtypes = [vp, ci, ci, ci, ci, ci, ci, ctypes.POINTER(vp)]
  L.dmc_batch_destroy.argtypes = [vp]
  L.dmc_batch_step.argtypes = [vp, ci, ci, vp]
  L.dmc_batch_forward.argtypes = [vp, ci, vp]
  L.dmc_batch_rollout.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp]
  L.dmc_batch_reset.argtypes = [vp, vp, ci]
  L.dmc_batch_field_rows.argtypes = [vp, cs, ctypes.POINTER(ci), ctypes.POINTER(ci)]
  L.dmc_batch_get.argtypes = [vp, cs, vp]
  L.dmc_batch_set.argtypes = [vp, cs, vp]
  L.dmc_batch_get_int.argtypes = [vp, cs, vp]
  L.dmc_batch_set_int.argtypes = [vp, cs, vp]
  L.dmc_batch_device_ptr.restype = vp
  L.dmc_batch_device_ptr.argtypes = [vp, cs]
  L.dmc_batch_bind.argtypes = [vp, cs, vp]
  L.dmc_batch_set_output_mask.argtypes = [vp, ci]
  L.dmc_batch_set_opt_int.argtypes = [vp, cs, ci]
  L.dmc_batch_set_opt_real.argtypes = [vp, cs, cd]
  L.dmc_batch_set_model_real.argtypes = [vp, cs, vp, ci]
  L.dmc_batch_sync.argtypes = [vp]
  L.dmc_batch_invalidate.argtypes = [vp]
  L.dmc_batch_invalidate_async.argtypes = [vp, vp]
  L.dmc_batch_step1.argtypes = [vp, vp]
  L.dmc_batch_create_caps.argtypes = [vp, ci, ci, ci, vp, ci, ctypes.POINTER(vp)]
  L.dmc_batch_set_env_geoms.argtypes = [vp, ci, vp]
  L.dmc_env_geom_pack.argtypes = [ci, vp, vp, vp, vp]
  L.dmc_gather_create.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_char_p), vp, vp, vp, ctypes.POINTER(vp)]
  L.dmc_gather_destroy.argtypes = [vp]
  L.dmc_gather_destroy.restype = None
  L.dmc_gather_run.argtypes = [vp, vp, vp]
  L.dmc_batch_step2.argtypes = [vp, vp]
  L.dmc_batch_info.argtypes = [vp, vp]
  L.dmc_batch_time_steps.argtypes = [vp, ci, ci, ci, vp, ctypes.POINTER(ctypes.c_float)]
  L.dmc_batch_enable_profiling.argtypes = [vp, ci]
  L.dmc_batch_set_step_probe.argtypes = [vp, ci, vp, ci]
  L.dmc_batch_set_async.argtypes = [vp, cs, vp, ci, vp]
  L.dmc_batch_get_async.argtypes = [vp, ci, ctypes.POINTER(cs), vp]
  L.dmc_batch_get_wait.argtypes = [vp, ci, ctypes.POINTER(vp), ci]
  L.dmc_batch_get_staged.argtypes = [vp, ci]
  L.dmc_batch_get_staged.restype = vp
  L.dmc_batch_get_timer.argtypes = [vp, ci, ctypes.POINTER(cd), ctypes.POINTER(ctypes.c_longlong)]
  L.dmc_batch_debug_enable.argtypes = [vp, ci]
  L.dmc_batch_debug_get.argtypes = [vp, cs, ci, vp, ctypes.POINTER(ci)]
  L.dmc_batch_prof_enable.argtypes = [vp, ci]
  L.dmc_batch_prof_get.argtypes = [vp, vp, ctypes.POINTER(ci)]
  L.dmc_batch_wave_trace.argtypes = [vp, ci, vp, ctypes.POINTER(ci)]
  L.dmc_batch_randomize_joints.argtypes = [vp, ctypes.c_uint64, vp, vp, ci, vp]
  if hasattr(L, 'dmc_batch_attach_specialised'):      # (absent from the older libraries A/B runs load as variants)
    L.dmc_batch_attach_specialised.argtypes = [vp, cs]
  if hasattr(L, 'dmc_batch_set_task_args'):
    L.dmc_batch_set_task_args.argtypes = [vp, vp, ci]
    L.dmc_batch_enable_task.argtypes = [vp, ci]
  _lib = L
  return L


def check(rc):
  if rc != 0:
    raise NativeError(lib().dmc_last_error().decode() or ('error %d' % rc))
