// emu.cpp -- TEST INFRASTRUCTURE ONLY.  Compiles dm_control_amd/csrc/step_core.h
// for the host with LPE = 1 (one "lane" per environment) so the kernel's
// indexing / algorithm logic can be unit-tested against the oracle without a
// GPU.  Nothing in the product imports or links this.
#define DMC_HOST_EMU 1
#include <cstdio>
#include <string>
#include <vector>

#include "../../dm_control_amd/csrc/step_core.h"
#include "../../dm_control_amd/csrc/step_tables.h"

using namespace dmc;

struct Emu {
  HostModel hm;
  StepTables tb;
  std::vector<float> mr32;
  // optional stash of the position / velocity stage between legacy steps (one environment)
  int stash_on = 0, stash_epoch = 1;
  std::vector<double> stash_r64; std::vector<float> stash_r32; std::vector<int> stash_i;
  // per-environment world geoms (one environment here): slot table + the 16 values per declared geom
  std::vector<int> eg_slot; std::vector<double> eg64; std::vector<float> eg32;
  std::vector<double> nsA;      // noslip matrix (global memory on the device)
  int kstash_on = 0; std::vector<double> kstash; std::vector<int> kstash_i;      // kinematic stash (doubles: room for either precision)
  std::vector<double> gs;       // the environment's global scratch (persists between launches, like the device buffer)
  std::vector<double> xfrc64; std::vector<float> xfrc32;      // xfrc_applied (6 nbody), empty = not set
  std::vector<double> mpos64, mquat64; std::vector<float> mpos32, mquat32;      // mocap_pos / mocap_quat (one environment)
};

static std::string g_err;

extern "C" {
const char* emu_last_error() { return g_err.c_str(); }
void* emu_create(const int32_t* ints, int ni, const double* reals, int nr, int nconmax, int njmax, int njcon) {
  Emu* e = new Emu;
  if (!host_model_parse(&e->hm, ints, ni, reals, nr, &g_err)) { delete e; return nullptr; }
  if (!step_tables_build(&e->tb, e->hm, nconmax, njmax, &g_err, njcon)) { delete e; return nullptr; }
  e->mr32.assign(e->tb.mr.begin(), e->tb.mr.end());
  return e;
}
void emu_free(void* h) { delete (Emu*)h; }
int emu_dims(void* h, int* out) {
  Emu* e = (Emu*)h; const StepLayout& L = e->tb.L;
  out[0] = L.n_sr + L.n_gs; out[1] = L.n_si; out[2] = L.d.nconmax; out[3] = L.d.njmax; out[4] = L.n_mi; out[5] = L.n_mr; out[6] = L.d.kmax; out[7] = L.n_mc;
  return 0;
}
// StepDims::treemax models: {treemax, ntreetri, ntree} and the tables dof_tree0 / dof_tree1 (nv each), tree_tri / tree_trim
int emu_tree_tables(void* h, int* dims, int* t0, int* t1, int* tri, int* trim) {
  Emu* e = (Emu*)h; const StepLayout& L = e->tb.L; const int* mi = e->tb.mi.data();
  dims[0] = L.d.treemax; dims[1] = L.d.ntreetri; dims[2] = L.d.ntree;
  if (L.d.treemax) {
    for (int i = 0; i < L.d.nv; i++) { t0[i] = mi[L.mi_dof_tree0 + i]; t1[i] = mi[L.mi_dof_tree1 + i]; }
    for (int k = 0; k < L.d.ntreetri; k++) { tri[k] = mi[L.mi_tree_tri + k]; trim[k] = mi[L.mi_tree_trim + k]; }
  }
  return 0;
}
int emu_split_solves_count() { return emu_split_solves(); }
long long emu_wsync_count() { return dmc_emu_wsync_count; }
void emu_ls_counts_get(long long* out) { for (int k = 0; k < 6; k++) out[k] = emu_ls_counts()[k]; }
void emu_stash(void* h, int on) { Emu* e = (Emu*)h; e->stash_on = on; e->stash_epoch++; e->stash_r64.assign(e->tb.L.n_keep + 4, 0.0); e->stash_r32.assign(e->tb.L.n_keep + 4, 0.f); e->stash_i.assign(e->tb.L.n_si + 4, 0); }
void emu_invalidate(void* h) { ((Emu*)h)->stash_epoch++; }
void emu_set_islands(void* h, int v) { ((Emu*)h)->tb.opts.islands = v; }      // StepOpts::islands: 1 on, 0 off, -1 by precision
void emu_kstash(void* h, int on) { Emu* e = (Emu*)h; const StepLayout& L = e->tb.L; e->kstash_on = on; e->kstash.assign(L.d.nq + L.d.nv + (L.s_qM - L.s_xpos) + 4, 0.0); e->kstash_i.assign(2, 0); }
void emu_set_xfrc(void* h, const double* x) { Emu* e = (Emu*)h; e->xfrc64.assign(x, x + 6*e->hm.nbody); e->xfrc32.assign(x, x + 6*e->hm.nbody); }
void emu_set_mocap(void* h, const double* p, const double* q) {
  Emu* e = (Emu*)h; const int n = e->hm.nmocap;
  e->mpos64.assign(p, p + 3*n); e->mpos32.assign(p, p + 3*n); e->mquat64.assign(q, q + 4*n); e->mquat32.assign(q, q + 4*n);
}
void emu_set_env_geoms(void* h, int n, const int* ids, const double* data) {
  Emu* e = (Emu*)h;
  e->eg_slot.assign(e->hm.ngeom, -1);
  for (int k = 0; k < n; k++) e->eg_slot[ids[k]] = k;
  e->eg64.assign(data, data + 16*n); e->eg32.assign(data, data + 16*n);
}
int emu_find(void* h, const char* name, int* off, int* cnt, int* kind) { return step_layout_find(&((Emu*)h)->tb.L, name, off, cnt, kind); }

}  // extern "C"

// io arrays are for ONE environment (B = 1).  prec: 64 or 32 (fp32 converts in/out).
template <typename T>
static void run_t(Emu* e, const T* mr, double** f, int** fi, int nstep, int legacy, int mode, double* dbg, int* dbgi) {
  const StepLayout& L = e->tb.L;
  StepOpts<T> o = step_opts_cast<T>(e->tb.opts);
  std::vector<T> s(L.n_sr, (T)0); std::vector<int> si(L.n_si, 0);
  // field order matches tests/emu_lib.py
  const int nb = L.d.nbody;
  const int sizes[] = {L.d.nq, L.d.nv, L.d.nu, L.d.nv, L.d.nv, 1,
                       L.d.nsensordata, 3*nb, 4*nb, 9*nb, 3*nb, 3*L.d.ngeom, 9*L.d.ngeom,
                       3*L.d.nsite, 9*L.d.nsite, 3*nb, L.d.nv, L.d.nu, L.d.nv, L.d.nv, L.d.nv,
                       L.d.nconmax, 3*L.d.nconmax, 9*L.d.nconmax, 6*L.d.nconmax, 6*nb, L.d.na};
  const int NF = sizeof(sizes)/sizeof(int);
  std::vector<std::vector<T>> buf(NF);
  for (int k = 0; k < NF; k++) { buf[k].resize(sizes[k] + 1); for (int i = 0; i < sizes[k]; i++) buf[k][i] = (T)f[k][i]; }
  std::vector<T> dbuf(L.n_sr + L.n_gs); std::vector<int> dibuf(L.n_si);
  StepIO<T> io;
  io.B = 1;
  io.qpos = buf[0].data(); io.qvel = buf[1].data(); io.ctrl = buf[2].data(); io.qacc_warmstart = buf[3].data();
  io.qfrc_applied = buf[4].data(); double tm = f[5][0]; io.time = &tm; io.prof = nullptr; io.ctrl_seq = nullptr; io.qpos_seq = nullptr; io.qvel_seq = nullptr; io.sensor_seq = nullptr;
  io.sensordata = buf[6].data(); io.xpos = buf[7].data(); io.xquat = buf[8].data(); io.xmat = buf[9].data();
  io.xipos = buf[10].data(); io.geom_xpos = buf[11].data(); io.geom_xmat = buf[12].data();
  io.site_xpos = buf[13].data(); io.site_xmat = buf[14].data(); io.subtree_com = buf[15].data();
  io.qacc = buf[16].data(); io.actuator_force = buf[17].data(); io.qfrc_actuator = buf[18].data();
  io.qfrc_bias = buf[19].data(); io.qfrc_constraint = buf[20].data();
  io.contact_dist = buf[21].data(); io.contact_pos = buf[22].data(); io.contact_frame = buf[23].data();
  io.contact_force = buf[24].data(); io.cvel = buf[25].data(); io.act = buf[26].data();
  io.ncon = fi[0]; io.nefc = fi[1]; io.solver_iter = fi[2]; io.warning = fi[3]; io.contact_geom1 = fi[4]; io.contact_geom2 = fi[5];
  io.debug = dbg ? dbuf.data() : nullptr; io.debug_i = dibuf.data(); io.ndebug = dbg ? 1 : 0;
  io.env_mode = nullptr; io.work = nullptr; io.cost = nullptr; io.order = nullptr; io.trace = nullptr; io.trace_slot = 0;
  if (!e->xfrc64.empty()) { o.xfrc = sizeof(T) == 8 ? (const void*)e->xfrc64.data() : (const void*)e->xfrc32.data(); o.xfrc_B = 1; }
  if (L.d.nmocap) {
    if (e->mpos64.empty()) {      // mj_resetData: the model poses
      std::vector<double> p(3*L.d.nmocap), q(4*L.d.nmocap);
      for (int i = 0; i < e->hm.nbody; i++) if (e->hm.body_mocapid[i] >= 0) { for (int k = 0; k < 3; k++) p[3*e->hm.body_mocapid[i] + k] = e->hm.body_pos[3*i + k]; for (int k = 0; k < 4; k++) q[4*e->hm.body_mocapid[i] + k] = e->hm.body_quat[4*i + k]; }
      emu_set_mocap(e, p.data(), q.data());
    }
    o.mocap_pos = sizeof(T) == 8 ? (const void*)e->mpos64.data() : (const void*)e->mpos32.data();
    o.mocap_quat = sizeof(T) == 8 ? (const void*)e->mquat64.data() : (const void*)e->mquat32.data(); o.mocap_B = 1;
  }
  o.g_mr = mr;
  e->gs.resize(L.n_gs + 2); o.gscr = e->gs.data();   // doubles: room for either precision
  if (L.d.nslip) { e->nsA.resize((size_t)L.d.nslip * L.d.nslip + 2); o.ns_A = e->nsA.data(); }
  if (!e->eg_slot.empty()) { o.eg_slot = e->eg_slot.data(); o.eg_n = (int)e->eg64.size() / 16; o.eg_B = 1; o.eg_data = sizeof(T) == 8 ? (const void*)e->eg64.data() : (const void*)e->eg32.data(); }
  io.stash_r = nullptr; io.stash_i = nullptr; io.epoch = &e->stash_epoch;
  io.kstash = e->kstash_on ? (T*)e->kstash.data() : nullptr; io.kstash_i = e->kstash_on ? e->kstash_i.data() : nullptr;
  if (e->stash_on) { io.stash_r = sizeof(T) == 8 ? (T*)e->stash_r64.data() : (T*)e->stash_r32.data(); io.stash_i = e->stash_i.data(); }
  DynLayoutSrc ls; ls.p = &L;
  StepCore<T, 1> core(ls, o, e->tb.mi.data(), mr, e->tb.mc.data(), s.data(), si.data(), 0);
  core.run(io, 0, nstep, legacy, mode, OUT_ALL, 1);
  for (int k = 0; k < NF; k++) for (int i = 0; i < sizes[k]; i++) f[k][i] = (double)buf[k][i];
  f[5][0] = tm;
  if (dbg) { for (int i = 0; i < L.n_sr + L.n_gs; i++) dbg[i] = (double)dbuf[i]; for (int i = 0; i < L.n_si; i++) dbgi[i] = dibuf[i]; }
}
extern "C" {
int emu_run(void* h, int prec, double** f, int** fi, int nstep, int legacy, int mode, double* dbg, int* dbgi) {
  Emu* e = (Emu*)h;
  if (prec == 64) run_t<double>(e, e->tb.mr.data(), f, fi, nstep, legacy, mode, dbg, dbgi);
  else run_t<float>(e, e->mr32.data(), f, fi, nstep, legacy, mode, dbg, dbgi);
  return 0;
}
}
