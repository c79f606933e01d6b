"""Model-specialised step kernels on demand (include/dmc_batch.h: dmc_batch_attach_specialised).

The library bakes specialised kernels for eight assets (build.py `_STATIC_MODELS`).  For any other model this module
compiles `csrc/step_kernel_spec.hip` with that model's LDS layout as a compile-time constant -- the layout comes from the
same host code the library uses (`gen_static_layouts`), the flags are those of the library unit a baked twin would live in
-- into a shared object cached on disk under the hash of (model blob, caps, precision, lanes, sources), and attaches it to
the batch.  A cold build is one hipcc run (about a minute for a small model, a few for a 60-dof one); a warm cache is a
dlopen.

  DMC_SPECIALISE = background (default): attach when the object is in the cache; otherwise compile it in a background thread
                                     (the batch runs the generic kernel meanwhile -- 2 - 3 x slower, logged once) and
                                     switch over at the first launch after the build has finished
                                     (`BatchedPhysics.wait_specialised()` blocks for it: callers that capture their
                                     launches into a HIP graph do that first)
                   cached:           attach when the object is already in the cache, never compile implicitly
                   build | 1:        compile at batch creation when it is missing (blocks for the build)
                   0:                never
  DMC_SPEC_CACHE:  cache directory (default: dm_control_amd/_spec_cache, in-tree so that it travels with the checkout)
  DMC_SPEC_PLUGIN: attach THIS object whatever the cache key says (tuning studies: an object built from an earlier state of
                   the sources against the current one, on one box)
  DMC_SPEC_FLAGS:  extra hipcc flags (tuning studies: `-DDMC_NO_...` switches of step_core.h; part of the cache key, so two
                   settings are two objects -- an A/B of one source tree on one box, each variant a 15 s build instead of
                   the library's five minutes: scripts/spec_variants.py)

`warm(model, precision=32, **caps)` builds ahead of time (e.g. once per composer model); `BatchedPhysics(...,
specialise=...)` overrides the environment per batch.  Batches whose layout equals a baked one keep the baked kernel.
"""
import concurrent.futures
import hashlib
import logging
import os
import shlex
import shutil
import subprocess
import tempfile
import threading

import numpy as np

from dm_control_amd import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
_SOURCES = ('step_core.h', 'step_layout.h', 'step_tables.h', 'step_kernel.hip.h', 'step_kernel_spec.hip',
            '../../include/dmc_model_layout.h', '../../include/dmc_batch.h')
_source_hash = None


def cache_dir():
  d = os.environ.get('DMC_SPEC_CACHE') or os.path.join(HERE, '_spec_cache')
  os.makedirs(d, exist_ok=True)
  return d


_log = logging.getLogger('dm_control_amd.specialise')
_toolchain = None
_pool = None
_pool_lock = threading.Lock()
_tool_lock = threading.Lock()
_inflight = {}      # plugin path -> Future of the background build producing it


def mode():
  m = os.environ.get('DMC_SPECIALISE', 'background').lower()
  return {'1': 'build', 'true': 'build', 'build': 'build', '0': 'off', 'off': 'off', 'false': 'off',
          'cached': 'cached'}.get(m, 'background')


def toolchain_id():
  """What besides the sources decides the object: the offload architecture and the compiler (the cache is in-tree and
  travels with the checkout -- possibly to a box with another ROCm).  Empty compiler part where hipcc is absent: such a
  box can only use objects built elsewhere with the same image."""
  global _toolchain
  if _toolchain is None:
    ver = ''
    try:
      ver = subprocess.run([_build.HIPCC, '--version'], capture_output=True, text=True, timeout=60).stdout
    except Exception:      # pylint: disable=broad-except
      pass
    keep = [l.strip() for l in ver.splitlines() if l.startswith(('HIP version', 'AMD clang version', 'clang version'))]
    _toolchain = _build.ARCH + '|' + '|'.join(keep)
  return _toolchain


def source_hash():
  global _source_hash
  if _source_hash is None:
    h = hashlib.sha1()
    for f in _SOURCES:
      with open(os.path.join(CSRC, f), 'rb') as fh:
        h.update(fh.read())
    _source_hash = h.hexdigest()
  return _source_hash


def _lean(model):
  """The small models' baked kernels leave the optional launch features out (step_core.h kFeat): so does their twin."""
  return model.nv < 30


def key(model, precision, lpe, caps, task_header=None):
  ints, reals = model.pack()
  h = hashlib.sha1()
  if task_header:      # a kernel with a task epilogue (suite/fused_env.py): the generated header is part of the object
    with open(task_header, 'rb') as fh:
      h.update(b'task:' + fh.read())
  h.update(ints.tobytes()); h.update(reals.tobytes())
  h.update(repr((int(precision), int(lpe), tuple(int(c) for c in caps), _lean(model), source_hash(), toolchain_id())).encode())
  h.update(os.environ.get('DMC_SPEC_FLAGS', '').encode())
  return h.hexdigest()[:24]


def path_for(model, precision, lpe, caps, task_header=None):
  return os.path.join(cache_dir(), 'libdmc_spec_%s.so' % key(model, precision, lpe, caps, task_header))


def build(model, precision=32, lpe=None, caps=(0, 0, 0), verbose=False, task_header=None):
  """Compiles the plugin for (model, caps = (nconmax, njmax, njcon) as given to the batch, precision, lanes); returns its
  path.  No-op when it is cached.  task_header: a task layer generated by suite/fused_env.py, evaluated by the kernel
  itself at the end of every step launch (the kernel then holds the optional launch features too)."""
  lpe = int(lpe or (32 if model.nv <= 12 else 64))
  out = path_for(model, precision, lpe, caps, task_header)
  if os.path.exists(out):
    return out
  # (-DDMC_PROFILE=1 among DMC_SPEC_FLAGS: the layout with the phase counters, for batches of libdmc_hip_prof.so)
  prof = 'DMC_PROFILE' in os.environ.get('DMC_SPEC_FLAGS', '')
  with _tool_lock:
    _build.generate_static_layouts(profile=prof)      # (makes sure the layout tool is built)
  tool = os.path.join(CSRC, 'gen_static_layouts' + ('_prof' if prof else ''))
  ints, reals = model.pack()
  with tempfile.TemporaryDirectory() as td:
    fi, fr = os.path.join(td, 'i.bin'), os.path.join(td, 'r.bin')
    ints.tofile(fi)
    reals.tofile(fr)
    init = subprocess.check_output([tool, 'spec', fi, fr] + [str(int(c)) for c in caps]).decode().strip()
    hdr = os.path.join(td, 'spec_layout.gen.h')
    with open(hdr, 'w') as f:
      f.write('\n'.join([
          '// GENERATED by dm_control_amd/specialise.py', '#pragma once',
          '#define DMC_STATIC_LAYOUT_0 %s' % init, '#define DMC_STATIC_NV_0 %d' % model.nv, '#define DMC_NSTATIC 1',
          '#define DMC_STATIC_LAYOUT_LIST {DMC_STATIC_LAYOUT_0}', '#define DMC_STATIC_INSTANCES(X) X(0, %d)' % lpe,
          '#define DMC_STATIC_INSTANCES_STD(X)', '#define DMC_STATIC_INSTANCES_ILP(X) X(0, %d)' % lpe,
          '#define DMC_STATIC_IDS(X) X(0)', '']))
    flags = list(_build._COMMON)      # pylint: disable=protected-access
    if precision == 64:
      flags += ['-ffp-contract=off']
    else:
      flags += ['-fno-hip-fp32-correctly-rounded-divide-sqrt', '-fgpu-flush-denormals-to-zero']
      if not _lean(model):
        flags += ['-mllvm', '-amdgpu-sched-strategy=max-ilp']      # as build.py's unit of the large models
    flags += ['-DDMC_LAYOUTS_HEADER="%s"' % hdr, '-DDMC_SPEC_PRECISION=%d' % precision, '-DDMC_SPEC_LPE=%d' % lpe,
              '-DDMC_STATIC_FEATURES=%d' % (0 if (_lean(model) and precision == 32) else 1)]
    if task_header:
      flags += ['-DDMC_TASK_HEADER="%s"' % os.path.abspath(task_header)]
    flags += shlex.split(os.environ.get('DMC_SPEC_FLAGS', ''))
    tmp = out + '.%d.tmp' % os.getpid()
    cmd = [_build.HIPCC] + flags + ['-shared', '-o', tmp, os.path.join(CSRC, 'step_kernel_spec.hip')]
    if verbose:
      print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, out)      # (atomic: concurrent builders of the same model race to the same content)
  return out


def warm(model, precision=32, lanes_per_env=0, nconmax=0, njmax=0, njcon=0, verbose=False, jlevel=None):
  """jlevel = 0: the kernel of a batch that keeps everything in LDS (17 .. 32 dofs and at most one environment per CU)."""
  caps = (nconmax, njmax, njcon) + ((jlevel + 1,) if jlevel is not None else ())
  return build(model, precision, lanes_per_env or None, caps, verbose)


def _describe(batch):
  m = batch.model
  return '%s (nv %d, B %d, fp%d)' % (getattr(m, 'model_name', None) or 'model', m.nv, batch.batch_size, batch.precision)


def _try_attach(batch, path):
  """dmc_batch_attach_specialised, without taking the batch down: an object the library refuses (built for another
  layout level, profile build, another model behind DMC_SPEC_PLUGIN, a foreign architecture in a copied cache) is logged
  and the batch keeps the generic kernel."""
  from dm_control_amd import _native
  rc = _native.lib().dmc_batch_attach_specialised(batch._ptr, path.encode())      # pylint: disable=protected-access
  if rc != 0:
    _log.warning('dm_control_amd: %s keeps the generic step kernel: %s was refused (%s)', _describe(batch), path,
                 _native.lib().dmc_last_error().decode(errors='replace'))
    return False
  return True


def build_async(model, precision, lpe, caps):
  """The plugin's build as a Future (one per object path, at most two hipcc runs at a time).  The workers are daemon
  threads: a process that exits while a build is running does not wait for it (the object appears under its final name
  only when complete)."""
  global _pool
  out = path_for(model, precision, lpe, caps)
  with _pool_lock:
    f = _inflight.get(out)
    if f is not None and not (f.done() and f.exception() is not None):
      return f
    if _pool is None:
      _pool = threading.Semaphore(2)
    f = concurrent.futures.Future()
    _inflight[out] = f

  def work():
    with _pool:
      try:
        f.set_result(build(model, precision, lpe, caps))
      except BaseException as ex:      # pylint: disable=broad-except
        f.set_exception(ex)
  threading.Thread(target=work, name='dmc-specialise', daemon=True).start()
  return f


def attach(batch, how=None):
  """Attaches the model's plugin to a BatchedPhysics that has no baked kernel.  Returns 'baked', 'attached', 'building'
  (a background build was started: `poll` / `wait` switch the batch over), 'missing' (not cached and not asked to build,
  or the object was refused) or 'off'."""
  from dm_control_amd import _native
  how = how or mode()
  if how == 'off' or not hasattr(_native.lib(), 'dmc_batch_attach_specialised'):
    return 'off'
  info = batch.info()
  if info['static_id'] >= 0:
    return 'baked' if info['static_id'] < 1000 else 'attached'
  caps = tuple(getattr(batch, '_user_caps', (0, 0, 0)))
  if batch.model.nv > 16 and info['global_scratch_bytes_per_env'] == 0:
    caps = caps + (1,)      # the batch keeps everything in LDS (a small batch, or caps[4] / DMC_JLEVEL): so must its kernel
  p = os.environ.get('DMC_SPEC_PLUGIN') or path_for(batch.model, batch.precision, info['lanes_per_env'], caps)
  if not os.path.exists(p):
    if how == 'cached':
      _log.info('dm_control_amd: %s runs the generic step kernel (2 - 3 x slower than a specialised one): none is cached and '
                'DMC_SPECIALISE=cached never builds; specialise.warm(model) builds it', _describe(batch))
      return 'missing'
    if shutil.which(_build.HIPCC) is None and not os.path.exists(_build.HIPCC):
      _log.warning('dm_control_amd: %s runs the generic step kernel (2 - 3 x slower): no specialised kernel is cached and '
                   'hipcc is not available to build one', _describe(batch))
      return 'missing'
    if how == 'background':
      # (one environment stepped through the (MjModel, MjData) seam gains nothing a 15 - 25 s compile is worth)
      if batch.batch_size < int(os.environ.get('DMC_SPEC_MIN_BATCH', '16')):
        return 'missing'
      batch._spec_future = build_async(batch.model, batch.precision, info['lanes_per_env'], caps)      # pylint: disable=protected-access
      batch._spec_path = p      # pylint: disable=protected-access
      _log.warning('dm_control_amd: %s runs the generic step kernel (2 - 3 x slower) while its specialised kernel is being '
                   'compiled in the background (one hipcc run, cached in %s)', _describe(batch), cache_dir())
      return 'building'
    try:
      build(batch.model, batch.precision, info['lanes_per_env'], caps)
    except (subprocess.CalledProcessError, OSError) as ex:
      _log.warning('dm_control_amd: %s keeps the generic step kernel: the build of its specialised kernel failed (%r)',
                   _describe(batch), ex)
      return 'missing'
  return 'attached' if _try_attach(batch, p) else 'missing'


def poll(batch, block=False, timeout=None):
  """Switches a batch whose plugin was being built in the background over to it once the build has finished (called
  by BatchedPhysics before a launch: a dictionary lookup while nothing is pending).  Returns batch.specialised."""
  f = getattr(batch, '_spec_future', None)
  if f is None:
    return batch.specialised
  if not block and not f.done():
    return batch.specialised
  try:
    f.result(timeout=timeout)
    ok = _try_attach(batch, batch._spec_path)      # pylint: disable=protected-access
  except concurrent.futures.TimeoutError:
    return batch.specialised
  except Exception as ex:      # pylint: disable=broad-except
    _log.warning('dm_control_amd: %s keeps the generic step kernel: the background build failed (%r)', _describe(batch), ex)
    ok = False
  batch._spec_future = None      # pylint: disable=protected-access
  batch.specialised = 'attached' if ok else 'missing'
  if ok:
    _log.warning('dm_control_amd: %s switched to its specialised step kernel', _describe(batch))
  return batch.specialised


def attach_task(batch, task_header, verbose=False):
  """Builds (blocking; cached) and attaches this batch's specialised kernel WITH the task epilogue of `task_header` -- also
  for a model whose plain kernel is baked into the library.  Returns True, or False when the object could not be built or
  was refused (the caller keeps its separate task kernel)."""
  info = batch.info()
  caps = tuple(getattr(batch, '_user_caps', (0, 0, 0)))
  if batch.model.nv > 16 and info['global_scratch_bytes_per_env'] == 0:
    caps = caps + (1,)
  try:
    p = build(batch.model, batch.precision, info['lanes_per_env'], caps, verbose=verbose, task_header=task_header)
  except (subprocess.CalledProcessError, OSError) as ex:
    _log.warning('dm_control_amd: %s: the kernel with the task epilogue could not be built (%r)', _describe(batch), ex)
    return False
  batch._spec_future = None      # pylint: disable=protected-access  (a plain kernel still building in the background is not wanted any more)
  if not _try_attach(batch, p):
    return False
  batch.specialised = 'attached'
  return True
