// lds_report.cpp -- build-time / tuning tool (host only, g++): prints the LDS footprint of a
// model array by array (tables and per-environment scratch).  Input: <name> <ints.bin> <reals.bin>
// [nconmax njmax njcon].  scripts/lds_report.py drives it over the BASELINE models.
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
  if (argc < 4) { fprintf(stderr, "usage: %s name ints.bin reals.bin [nconmax njmax njcon]\n", argv[0]); return 2; }
  std::vector<char> bi = slurp(argv[2]), br = slurp(argv[3]);
  dmc::HostModel hm; std::string err;
  if (!dmc::host_model_parse(&hm, (const int32_t*)bi.data(), (int)(bi.size()/4), (const double*)br.data(), (int)(br.size()/8), &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  dmc::StepTables tb;
  const int nconmax = argc > 4 ? atoi(argv[4]) : 0, njmax = argc > 5 ? atoi(argv[5]) : 0, njcon = argc > 6 ? atoi(argv[6]) : 0;
  if (!dmc::step_tables_build(&tb, hm, nconmax, njmax, &err, njcon)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  const StepLayout& L = tb.L;
  const StepDims& d = L.d;
  printf("{\"name\": \"%s\", \"nq\": %d, \"nv\": %d, \"nu\": %d, \"nbody\": %d, \"njnt\": %d, \"ngeom\": %d, \"npair\": %d, \"nM\": %d, \"ntri\": %d, "
         "\"nconmax\": %d, \"njmax\": %d, \"nslip\": %d, \"max_contacts\": %d, \"max_rows\": %d, \"n_mi\": %d, \"n_mr\": %d, \"n_mr_lds\": %d, \"n_gs\": %d, \"n_sr\": %d, \"n_si\": %d,\n",
         argv[1], d.nq, d.nv, d.nu, d.nbody, d.njnt, d.ngeom, d.npair, d.nM, d.ntri, d.nconmax, d.njmax, d.nslip, tb.max_contacts, tb.max_rows,
         L.n_mi, L.n_mr, L.n_mr_lds, L.n_gs, L.n_sr, L.n_si);
  printf(" \"int_tables\": {");
  { bool first = true;
#define X(n, c) { int cnt = (c); if (cnt) { printf("%s\"%s\": %d", first ? "" : ", ", #n, cnt); first = false; } }
    STEP_MODEL_INT_TABLES(X)
    printf("},\n \"real_tables\": {"); first = true;
    STEP_MODEL_HOT_REAL_TABLES(X)
    if (d.jglobal < 2) { STEP_MODEL_COLD_REAL_TABLES(X) }
    printf("},\n \"scratch_real\": {"); first = true;
    STEP_SCRATCH_REAL(X)
    printf("},\n \"ovl_pos\": {"); first = true;
    STEP_SCRATCH_OVL_POS(X)
    printf("},\n \"ovl_vel\": {"); first = true;
    STEP_SCRATCH_OVL_VEL(X)
    printf("},\n \"ovl_sol\": {"); first = true;
    STEP_SCRATCH_OVL_SOL(X)
    printf("},\n \"scratch_int\": {"); first = true;
    STEP_SCRATCH_INT(X)
#undef X
    printf("}}\n");
  }
  return 0;
}
