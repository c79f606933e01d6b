"""TEST INFRASTRUCTURE: runs the reference's OWN unit-test files -- read unmodified from /root/reference -- against THIS
package's modules.  The test file's imports are satisfied by a synthetic module table:

  absl.testing.absltest / parameterized -> unittest (TestCase, main; `parameters` expands a method per case)
  mock                                  -> unittest.mock
  dm_env, dm_env.specs                  -> dm_control_amd.envs.dm_env_api
  dm_control.<module under test>        -> the module of dm_control_amd named in `modules`

No reference source is copied; tests skip where the reference tree is absent."""
import importlib.util
import io
import os
import sys
import types
import unittest
from unittest import mock as _mock

from ref_root import REF  # noqa: E402  (/root/reference/dm_control, or the staged copy on the GPU box)


def available():
  return os.path.isdir(REF)


class _ParamMeta(type):

  def __new__(mcs, name, bases, ns):
    new = dict(ns)
    for key, fn in ns.items():
      cases = getattr(fn, '_dmc_cases', None)
      if cases is None:
        continue
      del new[key]
      for i, case in enumerate(cases):
        def bind(fn=fn, case=case):
          if isinstance(case, dict):
            return lambda self: fn(self, **case)
          if isinstance(case, (tuple, list)):
            return lambda self: fn(self, *case)
          return lambda self: fn(self, case)
        new['%s_%d' % (key, i)] = bind()
    return super().__new__(mcs, name, bases, new)


class _TestCase(unittest.TestCase, metaclass=_ParamMeta):
  # the absltest assertions the reference's tests use beyond unittest's
  def assertLen(self, container, n, msg=None):
    self.assertEqual(len(container), n, msg)

  def assertEmpty(self, container, msg=None):
    self.assertEqual(len(container), 0, msg)

  def assertNotEmpty(self, container, msg=None):
    self.assertGreater(len(container), 0, msg)

  def assertSameElements(self, a, b, msg=None):
    self.assertEqual(set(a), set(b), msg)

  def assertRaisesWithLiteralMatch(self, exc, literal, *args, **kwargs):
    import re
    return self.assertRaisesRegex(exc, '^' + re.escape(literal) + '$', *args, **kwargs)

  def assertRaisesWithPredicateMatch(self, exc, predicate, *args, **kwargs):
    case = self

    class Ctx:
      def __enter__(self):
        return self

      def __exit__(self, et, ev, tb):
        case.assertIsNotNone(et, '%s not raised' % exc.__name__)
        if not issubclass(et, exc):
          return False
        case.assertTrue(predicate(ev), 'predicate rejected %r' % (ev,))
        return True
    return Ctx()

  def assertSameStructure(self, a, b, aname='a', bname='b', msg=None):
    def walk(x, y, path):
      if isinstance(x, dict) and isinstance(y, dict):
        self.assertEqual(set(x), set(y), '%s: keys differ' % path)
        for k in x:
          walk(x[k], y[k], '%s[%r]' % (path, k))
      elif isinstance(x, (list, tuple)) and isinstance(y, (list, tuple)):
        self.assertEqual(len(x), len(y), '%s: lengths differ' % path)
        for i, (u, v) in enumerate(zip(x, y)):
          walk(u, v, '%s[%d]' % (path, i))
      else:
        self.assertEqual(x, y, msg or '%s differs' % path)
    walk(a, b, aname)

  def assertContainsSubset(self, expected_subset, actual_set, msg=None):
    missing = set(expected_subset) - set(actual_set)
    self.assertFalse(missing, msg or 'missing elements: %r' % sorted(missing))

  def assertBetween(self, value, lo, hi, msg=None):
    self.assertTrue(lo <= value <= hi, msg or '%r not in [%r, %r]' % (value, lo, hi))

  def assertStartsWith(self, actual, prefix, msg=None):
    self.assertTrue(actual.startswith(prefix), msg)

  def assertSequenceStartsWith(self, prefix, whole, msg=None):
    self.assertEqual(list(prefix), list(whole)[:len(list(prefix))], msg)


def _parameters(*cases):
  if len(cases) == 1 and not isinstance(cases[0], (tuple, dict, str)) and hasattr(cases[0], '__iter__'):
    cases = tuple(cases[0])

  def deco(fn):
    fn._dmc_cases = cases
    return fn
  return deco


def _named_parameters(*cases):
  if len(cases) == 1 and not isinstance(cases[0], (tuple, dict, str)) and hasattr(cases[0], '__iter__'):
    cases = tuple(cases[0])
  stripped = []
  for c in cases:
    if isinstance(c, dict):
      stripped.append({k: v for k, v in c.items() if k != 'testcase_name'})
    else:
      stripped.append(tuple(c[1:]))      # (name, *args)
  return _parameters(*stripped) if len(stripped) != 1 else _parameters(stripped)


def _install(modules):
  saved = {}

  def put(name, mod):
    saved[name] = sys.modules.get(name)
    sys.modules[name] = mod
  absl, testing = types.ModuleType('absl'), types.ModuleType('absl.testing')
  absl.__path__, testing.__path__ = [], []
  absltest, parameterized = types.ModuleType('absl.testing.absltest'), types.ModuleType('absl.testing.parameterized')
  absltest.TestCase, absltest.main, absltest.mock = _TestCase, (lambda *a, **k: None), _mock
  import tempfile
  absltest.get_default_test_tmpdir = lambda: os.path.join(tempfile.gettempdir(), 'dmc_amd_absl_testing')
  absltest.unittest = unittest
  parameterized.TestCase, parameterized.parameters = _TestCase, _parameters
  parameterized.named_parameters = _named_parameters
  absl.testing, testing.absltest, testing.parameterized = testing, absltest, parameterized
  for n, m in (('absl', absl), ('absl.testing', testing), ('absl.testing.absltest', absltest),
               ('absl.testing.parameterized', parameterized), ('mock', _mock)):
    put(n, m)
  from dm_control_amd.envs import dm_env_api
  put('dm_env', dm_env_api)
  put('dm_env.specs', dm_env_api.specs)
  if modules is None:      # the caller has a whole `dm_control` tree in sys.modules already (reference_pymjcf.bind_engine)
    return saved
  root = types.ModuleType('dm_control')
  root.__path__ = []
  put('dm_control', root)
  for dotted, mod in modules.items():
    parts = dotted.split('.')
    for k in range(1, len(parts)):
      pkg = '.'.join(parts[:k])
      if pkg not in sys.modules or (pkg != 'dm_control' and pkg not in saved):
        if pkg != 'dm_control':
          p = types.ModuleType(pkg)
          p.__path__ = []
          put(pkg, p)
          setattr(sys.modules['.'.join(parts[:k - 1])], parts[k - 1], p)
    put(dotted, mod)
    setattr(sys.modules['.'.join(parts[:-1])], parts[-1], mod)
  return saved


def _restore(saved):
  for name, mod in saved.items():
    if mod is None:
      sys.modules.pop(name, None)
    else:
      sys.modules[name] = mod


def run(test_file, modules, skip=()):
  """Executes REF/<test_file> with `modules` ({'dm_control.rl.control': module, ...}; None = leave the `dm_control`
  already in sys.modules alone) standing in for the reference's
  and runs every TestCase in it (`skip`: 'Class.method' prefixes that need something outside this backend's scope).
  Returns (unittest result, text report)."""
  saved = _install(modules)
  try:
    name = 'dmc_amd_reftest_' + test_file.replace('/', '_')[:-3]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, test_file))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    suite = unittest.TestSuite()
    loader = unittest.TestLoader()
    for obj in vars(mod).values():
      if isinstance(obj, type) and issubclass(obj, unittest.TestCase) and obj is not _TestCase:
        for t in loader.loadTestsFromTestCase(obj):
          ident = '%s.%s' % (type(t).__name__, t._testMethodName)
          if not any(ident.startswith(s) for s in skip):
            suite.addTest(t)
    out = io.StringIO()
    result = unittest.TextTestRunner(stream=out, verbosity=0).run(suite)
    return result, out.getvalue()
  finally:
    _restore(saved)
