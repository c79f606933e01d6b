"""Parses include/dmc_model_layout.h so Python packs the model blob in exactly
the order the C side unpacks it (the header is the single source of truth)."""
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'dmc_model_layout.h')


def _macro_body(text, name):
  m = re.search(r'#define\s+' + name + r'\(X\)\s*\\\n((?:.*\\\n)*.*\n)', text)
  if not m:
    raise RuntimeError('macro %s not found in %s' % (name, HEADER_PATH))
  return m.group(1)


def _load():
  with open(HEADER_PATH) as f:
    text = f.read()
  consts = {}
  for m in re.finditer(r'#define\s+(DMC_\w+)\s+(0x[0-9A-Fa-f]+|[-+0-9.eE]+)\s', text):
    v = m.group(2)
    consts[m.group(1)] = int(v, 16) if v.startswith('0x') else (
        float(v) if any(c in v for c in '.eE') else int(v))
  for m in re.finditer(r'(DMC_[A-Z0-9_]+)\s*=\s*([^,}]+)[,}]', text):
    expr = m.group(2).strip()
    try:
      consts[m.group(1)] = int(eval(expr, {}, {}))  # e.g. "1 << 4"
    except Exception:  # pylint: disable=broad-except
      pass
  hdr_ints = re.findall(r'X\((\w+)\)', _macro_body(text, 'DMC_MODEL_HEADER_INTS'))
  hdr_reals = re.findall(r'X\((\w+)\)', _macro_body(text, 'DMC_MODEL_HEADER_REALS'))
  int_fields = re.findall(r'X\((\w+),\s*([^)]+)\)', _macro_body(text, 'DMC_MODEL_INT_FIELDS'))
  real_fields = re.findall(r'X\((\w+),\s*([^)]+)\)', _macro_body(text, 'DMC_MODEL_REAL_FIELDS'))
  return consts, hdr_ints, hdr_reals, int_fields, real_fields


CONSTS, HEADER_INTS, HEADER_REALS, INT_FIELDS, REAL_FIELDS = _load()


def field_count(expr, sizes):
  return int(eval(expr, {}, dict(sizes)))  # count expressions are e.g. "3*nbody"
