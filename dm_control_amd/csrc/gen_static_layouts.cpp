// gen_static_layouts.cpp -- build-time tool (host only, g++): prints the LDS
// layout of a model as a C initializer so that step_kernel.hip.h can bake it
// into model-specialised kernel instantiations (offsets become immediates,
// loop bounds constants).  Input: <name> <ints.bin> <reals.bin> [nconmax njmax njcon jlevel+1].
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "step_tables.h"

static std::vector<char> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  std::vector<char> b;
  char tmp[65536]; size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n);
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s name ints.bin reals.bin [nconmax njmax njcon jlevel+1]\n", argv[0]); return 2; }
  std::vector<char> bi = slurp(argv[2]), br = slurp(argv[3]);
  dmc::HostModel hm; std::string err;
  if (!dmc::host_model_parse(&hm, (const int32_t*)bi.data(), (int)(bi.size()/4), (const double*)br.data(), (int)(br.size()/8), &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  dmc::StepTables tb;
  const int nconmax = argc > 4 ? atoi(argv[4]) : 0, njmax = argc > 5 ? atoi(argv[5]) : 0, njcon = argc > 6 ? atoi(argv[6]) : 0;
  const int jlevel = argc > 7 ? atoi(argv[7]) - 1 : -1;      // (0 = the default of the model's size)
  if (!dmc::step_tables_build(&tb, hm, nconmax, njmax, &err, njcon, jlevel)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  static_assert(sizeof(StepLayout) % sizeof(int) == 0, "StepLayout must be all ints");
  const int* p = (const int*)&tb.L;
  const int n = (int)(sizeof(StepLayout) / sizeof(int));
  printf("{");
  for (int i = 0; i < n; i++) printf("%s%d", i ? ", " : "", p[i]);
  printf("}\n");
  return 0;
}
