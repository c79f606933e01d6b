"""Which device functions of a kernel unit differ between two builds: disassembles the code objects of
dm_control_amd/csrc/<unit>.o in two trees and compares each function's instruction text (addresses and branch-target
comments stripped).  Used to show that a change compiles out of the benchmarked kernels.

  python scripts/compare_kernel_code.py <other tree> [unit.o ...]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scratch_by_function import device_elf, demangle, OBJDUMP, CSRC, ROOT      # noqa: E402


def functions(obj):
  """function -> list of instruction lines"""
  with tempfile.TemporaryDirectory() as td:
    asm = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', '--no-leading-addr', device_elf(obj, td)],
                         capture_output=True, text=True).stdout
  out, cur = {}, None
  for line in asm.split('\n'):
    m = re.match(r'^(?:[0-9a-f]+ )?<(.+)>:', line)
    if m:
      cur = m.group(1)
      out[cur] = []
      continue
    if cur and line.strip():
      text = re.sub(r'//.*', '', line)
      text = re.sub(r'<[^>]*>', '', text)      # branch-target labels carry absolute offsets of the unit
      out[cur].append(text.strip())
  return out


if __name__ == '__main__':
  other = sys.argv[1]
  for unit in sys.argv[2:] or ['step_kernels_f32.o', 'step_kernels_f32_ilp.o']:
    a = functions(os.path.join(CSRC, unit))
    b = functions(os.path.join(other, 'dm_control_amd', 'csrc', unit))
    diff = sorted(k for k in set(a) | set(b) if a.get(k) != b.get(k))
    print('%s: %d functions, %d not byte-identical' % (unit, len(set(a) | set(b)), len(diff)))
    for name, short in zip(diff, demangle(diff)):
      if name not in a or name not in b:
        print('   %-60s only in one build' % short[:60])
      elif len(a[name]) != len(b[name]):
        print('   %-60s %d vs %d instructions' % (short[:60], len(a[name]), len(b[name])))
      else:
        # same length: which instructions differ?  `s_add_u32 sN, sN, imm` after s_getpc_b64 is the pc-relative address of a
        # callee or a constant (it moves when OTHER functions of the unit change size); v_mov / s_mov of a small immediate
        # is a byte offset into a struct that gained a member
        ops = {}
        for x, y in zip(a[name], b[name]):
          if x != y:
            ops[x.split()[0]] = ops.get(x.split()[0], 0) + 1
        print('   %-60s %d instructions, same opcodes; operands differ in: %s' % (short[:60], len(a[name]), ops))
