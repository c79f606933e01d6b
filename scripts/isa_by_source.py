"""ISA of one code object attributed to source functions of step_core.h (build the plugin with -gline-tables-only):
   python scripts/isa_by_source.py <plugin.so> <stage: posvel|acc|euler|kernel> [function ...]     -- listing of those functions' instructions
   without function names: static instruction counts per source function."""
import bisect, collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, 'dm_control_amd/csrc/step_core.h')).read().splitlines()
starts, names = [], []
for i, l in enumerate(src, 1):
  m = re.match(r'\s*(?:template <[^>]*>\s*)?(?:static )?(?:DMC_DEV|DMC_FN)\s+(?:static\s+)?[\w:<>\*& ,\(\)]*?\b(\w+)\s*\(', l)
  if m and not l.strip().startswith('//'):
    starts.append(i); names.append(m.group(1))
def fn(line):
  k = bisect.bisect_right(starts, line) - 1
  return names[k] if k >= 0 else '?'
blob = open(sys.argv[1], 'rb').read()
pos = blob.find(b'\x7fELF', 4)
td = tempfile.mkdtemp()
open(td + '/co.elf', 'wb').write(blob[pos:])
out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '-l', '--no-show-raw-insn', td + '/co.elf'], capture_output=True, text=True).stdout.splitlines()
stage, want = sys.argv[2], set(a for a in sys.argv[3:] if not a.startswith('@'))
span = [a[1:] for a in sys.argv[3:] if a.startswith('@')]      # @function: everything between its first and last instruction
rows = []
func = cur = None
cnt = collections.Counter()
for l in out:
  m = re.match(r'^[0-9a-f]+ <(.*)>:', l)
  if m:
    f = m.group(1)
    func = 'posvel' if 'posvel' in f else 'acc' if '3accE' in f else 'euler' if '5euler' in f else 'kernelQ' if 'Lb1EE' in f else 'kernel' if 'step_kernel' in f else f[:30]
    continue
  m = re.match(r'^; (.*):(\d+)$', l)
  if m:
    cur = (m.group(1).split('/')[-1], int(m.group(2))); continue
  if l.startswith('\t') and func == stage and cur:
    key = fn(cur[1]) if cur[0] == 'step_core.h' else cur[0]
    cnt[key] += 1
    rows.append((cur[1], key, re.sub(r'\s*//.*', '', l.strip())[:110]))
    if key in want:
      print('%-5d %-22s %s' % (cur[1], key[:22], re.sub(r'\s*//.*', '', l.strip())[:110]))
if span:
  idx = [k for k, r in enumerate(rows) if r[1] == span[0]]
  for r in rows[idx[0]:idx[-1] + 1]: print('%-5d %-22s %s' % (r[0], r[1][:22], r[2]))
elif not want:
  print(stage, sum(cnt.values()))
  for k, v in cnt.most_common(60): print('  %-28s %d' % (k, v))
