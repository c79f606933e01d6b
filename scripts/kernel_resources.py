"""Register / scratch / LDS use of every step-kernel instantiation, read from the code-object notes of the built objects
(no recompilation): VGPRs, spilled VGPRs, private segment (scratch) bytes per lane, SGPRs."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dm_control_amd', 'csrc')


def kernels(obj):
  blob = open(obj, 'rb').read()
  out = {}
  pos = blob.find(b'\x7fELF')
  with tempfile.TemporaryDirectory() as td:
    while pos >= 0:
      elf = os.path.join(td, 'o%d.elf' % pos)
      open(elf, 'wb').write(blob[pos:])
      r = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', elf], capture_output=True, text=True)
      if 'amdhsa.kernels' in r.stdout:
        for blk in r.stdout.split('- .agpr_count')[1:]:
          name = re.search(r'\.name:\s+(\S+)', blk).group(1)
          g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
          out[name] = dict(vgpr=g('vgpr_count'), vgpr_spill=g('vgpr_spill_count'), sgpr=g('sgpr_count'),
                           sgpr_spill=g('sgpr_spill_count'), scratch=g('private_segment_fixed_size'))
      pos = blob.find(b'\x7fELF', pos + 4)
  return out


def short(name):
  m = re.search(r'step_kernel(_static)?I([fd])Li(\d+)(?:ELi(\d+))?ELb([01])', name)
  if not m:
    return name[:40]
  return '%s<%s,%s%s>%s' % ('static' if m.group(1) else 'generic', 'f32' if m.group(2) == 'f' else 'f64', m.group(3),
                            (',' + m.group(4)) if m.group(4) else '', ' queue' if m.group(5) == '1' else '')


if __name__ == '__main__':
  for unit in sys.argv[1:] or ['step_kernels_f32.o', 'step_kernels_f32_ilp.o', 'step_kernels_f64.o']:
    ks = kernels(os.path.join(CSRC, unit))
    print(unit)
    for n in sorted(ks, key=short):
      print('  %-28s %s' % (short(n), ks[n]))
