"""Where a kernel unit's private-segment (scratch) accesses are: scratch_load / scratch_store instructions per function of
the built device code object, optionally with the source lines they come from.

  python scripts/scratch_by_function.py [unit.o]            # static counts per function (largest stack offset beside it)
  python scripts/scratch_by_function.py --lines <substring>  # recompiles the ilp unit with -gline-tables-only (codegen is
                                                            # unchanged) and lists (count, load/store, file:line) for the
                                                            # functions whose mangled name contains <substring>

Round 4 used it to find that the 388 B / lane of the CMU kernel were not stack arrays but the out-of-line stage functions
saving 90 callee-saved VGPRs at entry and restoring them at exit (step_core.h DMC_FN: `not_tail_called` lets LLVM's
interprocedural register allocation drop those saves)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dm_control_amd', 'csrc')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def device_elf(obj, td):
  blob = open(obj, 'rb').read()
  pos, k = blob.find(b'\x7fELF'), 0
  while pos >= 0:
    path = os.path.join(td, 'o%d.elf' % k)
    open(path, 'wb').write(blob[pos:])
    r = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '-h', path], capture_output=True, text=True)
    if 'AMDGPU' in r.stdout or 'AMD GPU' in r.stdout:
      return path
    pos, k = blob.find(b'\x7fELF', pos + 4), k + 1
  raise RuntimeError('no device code object in ' + obj)


def scan(obj, lines=False):
  with tempfile.TemporaryDirectory() as td:
    elf = device_elf(obj, td)
    asm = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn'] + (['-l'] if lines else []) + [elf],
                         capture_output=True, text=True).stdout
  cur, where = None, None
  count, offs = collections.defaultdict(collections.Counter), collections.defaultdict(int)
  for line in asm.split('\n'):
    m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
    if m:
      cur = m.group(1)
      continue
    m = re.match(r'^; (.+):(\d+)\s*$', line)
    if m:
      where = '%s:%s' % (os.path.basename(m.group(1)), m.group(2))
      continue
    if 'scratch_' in line:
      count[cur][(where if lines else None, 'load' if 'scratch_load' in line else 'store')] += 1
      mo = re.search(r'offset:(\d+)', line)
      offs[cur] = max(offs[cur], int(mo.group(1)) if mo else 0)
  return count, offs


def demangle(names):
  out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
  short = []
  for n in out:
    n = re.sub(r'dmc::StageFns<(\w+), (\d+), dmc::StaticLayout<(-?\d+)> ?>', r'StageFns<\1,\2,\3>', n)
    short.append(re.sub(r'\(.*', '', n))
  return short


if __name__ == '__main__':
  args = sys.argv[1:]
  if args and args[0] == '--lines':
    sub = args[1]
    with tempfile.TemporaryDirectory() as td:
      obj = os.path.join(td, 'ilp_g.o')
      subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                             '-fno-hip-fp32-correctly-rounded-divide-sqrt', '-fgpu-flush-denormals-to-zero', '-mllvm',
                             '-amdgpu-sched-strategy=max-ilp', '-gline-tables-only', '-c',
                             os.path.join(CSRC, 'step_kernels_f32_ilp.hip'), '-o', obj], stderr=subprocess.DEVNULL)
      count, _ = scan(obj, lines=True)
    for fn in count:
      if sub in fn:
        print(demangle([fn])[0])
        for (where, kind), n in sorted(count[fn].items(), key=lambda kv: (kv[0][0] or '', kv[0][1])):
          print('   %4d %-5s %s' % (n, kind, where))
  else:
    for unit in args or ['step_kernels_f32.o', 'step_kernels_f32_ilp.o', 'step_kernels_f64.o']:
      count, offs = scan(os.path.join(CSRC, unit))
      fns = sorted(count, key=lambda f: -sum(count[f].values()))
      print(unit)
      for fn, name in zip(fns, demangle(fns)):
        print('   %4d instr  max offset %4d  %s' % (sum(count[fn].values()), offs[fn], name[:110]))
