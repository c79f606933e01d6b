// dmc_api.hip -- host side of libdmc_hip.so: the C-ABI of include/dmc_batch.h.
// Owns device memory (SoA mjData arrays), builds the kernel tables / LDS layout
// and launches the fused step kernel.  No CPU fallback exists: every entry point
// that computes runs the HIP kernel or returns an error.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dmc_batch.h"
#include "step_kernel.hip.h"
#include "step_tables.h"

namespace dmc {
hipError_t launch_step_f32(const LaunchGeom& g, hipStream_t stream, const StepLayout* d_layout, const StepOpts<float>& o,
                           const int* g_mi, const float* g_mr, const int* g_mc, const StepIO<float>& io, int nstep, int legacy, int mode, int outmask, int nsub);
hipError_t launch_step_f64(const LaunchGeom& g, hipStream_t stream, const StepLayout* d_layout, const StepOpts<double>& o,
                           const int* g_mi, const double* g_mr, const int* g_mc, const StepIO<double>& io, int nstep, int legacy, int mode, int outmask, int nsub);
}  // namespace dmc

using namespace dmc;

static thread_local std::string g_err;
static int fail(const std::string& msg, int code = -1) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_), -2); \
  } while (0)

struct dmc_model {
  HostModel hm;
};

struct Field {
  std::string name;
  int rows;
  bool is_int;
  bool is_f64;     // stored as double regardless of the batch precision (time)
  void* dev;       // current binding
  void* owned;     // hipMalloc'ed by us (may differ from dev after dmc_batch_bind)
};

struct dmc_batch {
  const dmc_model* model;
  int B, device, precision;
  StepTables tb;
  LaunchGeom geom;
  int* d_mi;
  int* d_mc;      // cold int tables (stay in global memory)
  void* d_mr;
  StepLayout* d_layout;
  std::vector<Field> fields;
  std::map<std::string, int> index;
  int outmask;
  int ndebug;
  void* d_debug;
  int* d_debug_i;
  long long* d_prof;
  size_t elem;  // sizeof(T)
  // per-env stash of the position / velocity stage between legacy steps (StepIO::stash_*); epoch: bumped by every
  // host-side edit that can change what the stage depends on, which invalidates all stashes at once
  void* d_stash_r; int* d_stash_i; int* d_epoch; int stash_on; int stash_auto;
  int xfrc_on;         // xfrc_applied was written / bound / exposed: the kernel reads it from now on
  void* d_ns_A;        // noslip: (B, nslip, nslip) reals in global memory (StepOpts::ns_A)
  int* d_work;         // work queue of launches with a resident-only grid: {next item, finished waves} (StepIO::work)
  int ncu;             // compute units of the device
  void* d_kstash; int* d_kstash_i;      // kinematic stash (StepIO::kstash), on unless DMC_NO_KSTASH
  int *d_cost, *d_order; int lpt, nitems;      // longest-first scheduling of queued launches (StepIO::cost / order)
  int* d_prog = nullptr; int max_slices = 0;   // sliced items of queued multi-step launches (StepIO::prog / slices)
  unsigned long long* d_hand = nullptr; int hand_n = 0; int nxcd = 1;      // hand-off records of the pieces; queues per launch (one per XCD)
  void* d_gscr;        // large models: (B, n_gs) reals of per-env global scratch (StepOpts::gscr)
  int* d_trace;        // wave trace (dmc_batch_wave_trace): ring of 8 launches x (8, nitems) ints, or null
  struct Profiler* prof = nullptr;      // launch timers (dmc_batch_enable_profiling), or null
  struct Xfer* xfer = nullptr;      // pinned / device staging of the asynchronous host transfers (dmc_batch_set_async / get_async), or null
  void* d_probe = nullptr; int probe_geom = 0, probe_cap = 0;      // substep probe (dmc_batch_set_step_probe): caller-owned (cap, 3, B) reals
  int trace_launch;    // launches since the trace was switched on (ring slot = trace_launch % 8)
  int* d_rj_i; double* d_rj_r;      // joint randomisation: (4, njnt) ints {type, qposadr, limited, 0} and (2, njnt) ranges
  int* d_eg_slot;      // per-env world geoms: (ngeom) slot table on the device (field "env_geom" holds the values)
  // a model-specialised kernel built on demand for this model (dmc_batch_attach_specialised): the loaded object, its launch
  // entry, whether it holds the optional launch features (step_core.h kFeat)
  void* spec_so = nullptr; void* spec_launch = nullptr; int spec_features = 0;
  int spec_task_bytes = 0;      // > 0: the attached kernel carries a task epilogue with an argument block of this size
  void* d_task_args = nullptr; int task_on = 0;      // its arguments on the device; whether the next step launches run it
};

extern "C" const char* dmc_last_error(void) { return g_err.c_str(); }

extern "C" int dmc_model_create(const int32_t* ints, int n_ints, const double* reals, int n_reals, dmc_model** out) {
  if (!ints || !reals || !out) return fail("null argument");
  dmc_model* m = new dmc_model;
  std::string err;
  if (!host_model_parse(&m->hm, ints, n_ints, reals, n_reals, &err)) { delete m; return fail(err); }
  *out = m;
  return 0;
}
extern "C" void dmc_model_destroy(dmc_model* m) { delete m; }

static Field* find_field(dmc_batch* b, const char* name) {
  auto it = b->index.find(name);
  return it == b->index.end() ? nullptr : &b->fields[it->second];
}

static int choose_geometry(dmc_batch* b, int lanes_per_env) {
  const StepLayout& L = b->tb.L;
  const size_t tables = (size_t)(b->precision == 64 ? opts_lds_bytes<double>() : opts_lds_bytes<float>()) +
                        (size_t)L.n_mi * sizeof(int) + (size_t)L.n_mr_lds * b->elem + (L.d.coldlds ? (size_t)L.n_mc * sizeof(int) : 0);
  const size_t env_bytes = (size_t)L.n_sr * b->elem + (size_t)L.n_si * sizeof(int);
  const size_t lds_cu = 160 * 1024;
  // automatic: small models (cheetah, nv = 9) leave most of a 64-lane group idle, so
  // two environments share a wavefront; measured on MI355X (scripts/perf_probe.py)
  int lpe = lanes_per_env ? lanes_per_env : (b->tb.L.d.nv <= 12 ? 32 : 64);
  if (lpe != 64 && lpe != 32 && lpe != 16) return fail("lanes_per_env must be 64, 32 or 16");
  const int epw = 64 / lpe;
  int best_w = 0; long best_score = -1, best_blocks = 1;
  // The kernels are built for 2 waves per SIMD (256 VGPRs): at most 8 resident waves per CU whatever the LDS would
  // allow.  Workgroups of more than 4 waves were tried (512 threads: humanoid 7 environments in one workgroup instead
  // of 2 x 3) and measured slower (1.20 M vs 1.29 M env-steps/s), so 4 waves stays the largest shape.
  const int force_w = getenv("DMC_WAVES") ? atoi(getenv("DMC_WAVES")) : 0;      // tuning studies only
  // ... with one exception: FIVE waves when that is what holds one more environment than any smaller shape (the 62-dof
  // models at offload level 3: 5 x 29.8 KB + one 10.8 KB table copy = 159.6 KB; +25 % residency, round 4)
  for (int w = force_w > 5 ? force_w : 5; w >= 1; w--) {      // (more than five waves only when forced: needs kernels built with -DDMC_MAX_THREADS=<64 w>)
    if (force_w && w != force_w) continue;
    const size_t bytes = tables + (size_t)w * epw * env_bytes;
    if (bytes > lds_cu) continue;
    long blocks = (long)(lds_cu / bytes);
    if (blocks * w > 8) blocks = 8 / w;             // VGPR-limited: 2 waves per SIMD
    if (blocks < 1) continue;
    const long score = blocks * w * epw;            // resident envs per CU
    // fewer waves per workgroup only for a clear gain in residency: 4-wave groups give grids that divide the
    // batch evenly (cartpole, B = 4096: 3-wave groups = 683 workgroups ran 1.8x slower than 4-wave = 512)
    if (best_score < 0 || score * 100 > best_score * 115) { best_score = score; best_w = w; best_blocks = blocks; }
  }
  // a batch too small to fill the chip (soccer: 256 environments per GPU): spread it over as many CUs as possible --
  // the smallest workgroup that still holds the batch in one round (one wave alone on a CU shares nothing)
  if (best_w) {
    const long cus = b->ncu > 0 ? b->ncu : 256, need = ((long)b->B + cus * epw - 1) / (cus * epw);      // waves per CU to hold B at once
    if (need < best_w && !force_w) { best_w = (int)std::max(1L, need); best_blocks = std::min(8L / best_w, (long)(lds_cu / (tables + (size_t)best_w * epw * env_bytes))); }
  }
  if (!best_w) return fail("environment scratch does not fit in 160 KiB of LDS; lower nconmax/njmax");
  LaunchGeom& g = b->geom;
  g.lpe = lpe; g.waves = best_w; g.envs_per_block = best_w * epw;
  g.lds_bytes = (int)(tables + (size_t)g.envs_per_block * env_bytes);
  g.grid = (b->B + g.envs_per_block - 1) / g.envs_per_block;
  g.blocks_per_cu = (int)std::max(1L, best_blocks);
  // a batch of more workgroups than the chip holds at once: launch the resident ones, their waves claim the rest
  // from the work queue as they finish (step_kernel_body)
  g.queue = 0;
  if (g.grid > b->ncu * g.blocks_per_cu && !getenv("DMC_NO_QUEUE")) { g.grid = b->ncu * g.blocks_per_cu; g.queue = 1; }
  g.static_id = -1;
#if DMC_NSTATIC > 0
  {
    static const StepLayout baked[DMC_NSTATIC] = DMC_STATIC_LAYOUT_LIST;
    for (int k = 0; k < DMC_NSTATIC; k++) if (!memcmp(&baked[k], &L, sizeof(StepLayout))) g.static_id = k;
    if (getenv("DMC_NO_STATIC")) g.static_id = -1;
  }
#endif
  return 0;
}

static int upload_tables(dmc_batch* b) {
  const StepLayout& L = b->tb.L;
  HIP_TRY(hipMemcpy(b->d_mi, b->tb.mi.data(), (size_t)L.n_mi * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_mc, b->tb.mc.data(), b->tb.mc.size() * sizeof(int), hipMemcpyHostToDevice));
  if (b->precision == 64) {
    HIP_TRY(hipMemcpy(b->d_mr, b->tb.mr.data(), (size_t)L.n_mr * sizeof(double), hipMemcpyHostToDevice));
  } else {
    std::vector<float> f(b->tb.mr.begin(), b->tb.mr.end());
    HIP_TRY(hipMemcpy(b->d_mr, f.data(), (size_t)L.n_mr * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int dmc_batch_create(const dmc_model* m, int batch_size, int device_id, int precision,
                                int nconmax, int njmax, int lanes_per_env, dmc_batch** out) {
  const int caps[3] = {nconmax, njmax, lanes_per_env};
  return dmc_batch_create_caps(m, batch_size, device_id, precision, caps, 3, out);
}
extern "C" int dmc_batch_create_caps(const dmc_model* m, int batch_size, int device_id, int precision,
                                     const int* caps, int ncaps, dmc_batch** out) {
  if (ncaps < 0 || (ncaps > 0 && !caps)) return fail("null argument");
  const int nconmax = ncaps > 0 ? caps[0] : 0, njmax = ncaps > 1 ? caps[1] : 0, lanes_per_env = ncaps > 2 ? caps[2] : 0;
  const int njcon = ncaps > 3 ? caps[3] : 0;
  int jlevel = ncaps > 4 ? caps[4] - 1 : -1;      // caps[4]: StepDims::jglobal + 1, 0 = automatic
  if (!m || !out) return fail("null argument");
  if (batch_size < 1) return fail("batch_size must be >= 1");
  if (precision != 32 && precision != 64) return fail("precision must be 32 or 64");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device available: the batched step has no CPU fallback", -3);
  if (device_id < 0 || device_id >= ndev) return fail("invalid device id");
  HIP_TRY(hipSetDevice(device_id));
  dmc_batch* b = new dmc_batch();
  b->model = m; b->B = batch_size; b->device = device_id; b->precision = precision;
  b->elem = precision == 64 ? sizeof(double) : sizeof(float);
  b->outmask = OUT_ALL; b->ndebug = 0; b->d_debug = nullptr; b->d_debug_i = nullptr; b->d_mi = nullptr; b->d_mc = nullptr; b->d_mr = nullptr; b->d_stash_r = nullptr; b->d_stash_i = nullptr; b->d_epoch = nullptr; b->stash_on = 0; b->stash_auto = 0; b->d_eg_slot = nullptr; b->xfrc_on = 0; b->d_ns_A = nullptr; b->d_gscr = nullptr; b->d_work = nullptr; b->d_kstash = nullptr; b->d_kstash_i = nullptr; b->d_cost = nullptr; b->d_order = nullptr; b->lpt = 0; b->nitems = 0; b->d_prof = nullptr; b->d_layout = nullptr; b->d_trace = nullptr; b->trace_launch = 0; b->d_rj_i = nullptr; b->d_rj_r = nullptr;
  std::string err;
  { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || ncu < 1) ncu = 256; b->ncu = ncu; }
  // The contact rows and the kept factor of M leave LDS for the per-env global scratch on the 17 .. 32-dof models so that
  // more environments are resident per CU (level 1: config 3 +24 %).  A batch of at most one environment per CU has
  // nothing to gain from residency and pays a global round trip wherever a row is read: it keeps them in LDS
  // (soccer 2v2, B = 256: +3 %; at B = 4096 the same choice costs 8 %).  DMC_JLEVEL=<n>: tuning override.
  if (jlevel < 0 && getenv("DMC_JLEVEL")) jlevel = atoi(getenv("DMC_JLEVEL"));
  const bool small_auto = jlevel < 0 && batch_size <= b->ncu && DMC_JGLOBAL_LEVEL(m->hm.nv) == 1;
  if (!step_tables_build(&b->tb, m->hm, nconmax, njmax, &err, njcon, small_auto ? 0 : jlevel)) { delete b; return fail(err); }
  if (small_auto) {
    // ... unless only the default layout has a baked model-specialised kernel (the generic one is 2 - 3 x slower: that
    // would be a bad trade for a global round trip per row)
    // ... and unless the level-0 layout does not fit 160 KiB of LDS at all (a large caller-chosen nconmax / njmax): the
    // default level, which keeps those rows out of LDS, is what such a batch was built with before this choice existed
    const bool fits = choose_geometry(b, lanes_per_env) == 0;
    if (!fits || b->geom.static_id < 0) {
      StepTables def;
      if (step_tables_build(&def, m->hm, nconmax, njmax, &err, njcon, -1)) {
        StepTables keep = b->tb; LaunchGeom kg = b->geom;
        b->tb = def;
        const bool def_fits = choose_geometry(b, lanes_per_env) == 0;
        if (fits && (!def_fits || b->geom.static_id < 0)) { b->tb = keep; b->geom = kg; }
      } else if (!fits) { delete b; return fail(err); }
    }
  }
  if (choose_geometry(b, lanes_per_env)) { delete b; return -1; }
  const StepLayout& L = b->tb.L;
  const StepDims& d = L.d;
  hipError_t e = hipMalloc(&b->d_mi, (size_t)L.n_mi * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&b->d_mc, b->tb.mc.size() * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&b->d_mr, (size_t)L.n_mr * b->elem);
  if (e == hipSuccess) e = hipMalloc((void**)&b->d_layout, sizeof(StepLayout));
  if (e == hipSuccess) e = hipMemcpy(b->d_layout, &L, sizeof(StepLayout), hipMemcpyHostToDevice);
  if (e != hipSuccess) { delete b; return fail(std::string("hipMalloc: ") + hipGetErrorString(e), -2); }
  if (upload_tables(b)) { delete b; return -2; }
  b->tb.opts.g_mr = b->d_mr;
  if (d.nslip) {
    e = hipMalloc(&b->d_ns_A, (size_t)b->B * d.nslip * d.nslip * b->elem);
    if (e != hipSuccess) { dmc_batch_destroy(b); return fail(std::string("hipMalloc noslip matrix: ") + hipGetErrorString(e), -2); }
    b->tb.opts.ns_A = b->d_ns_A;
  }
  { const int one = 1;
    e = hipMalloc((void**)&b->d_epoch, sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(b->d_epoch, &one, sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { dmc_batch_destroy(b); return fail(std::string("hipMalloc stash epoch: ") + hipGetErrorString(e), -2); } }
  // work[1]: finished waves; work[32 (1 + x)]: head of XCD x's queue (a 128-byte line each)
  e = hipMalloc((void**)&b->d_work, 32 * 9 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(b->d_work, 0, 32 * 9 * sizeof(int));
  if (e != hipSuccess) { dmc_batch_destroy(b); return fail(std::string("hipMalloc work queue: ") + hipGetErrorString(e), -2); }
  // (a mocap pose is an input of mj_kinematics that the stash's (qpos, qvel) comparison does not see: no stash for those models)
  if (!getenv("DMC_NO_KSTASH") && !d.nmocap) {
    const size_t nk = (size_t)d.nq + d.nv + (L.s_qM - L.s_xpos);
    e = hipMalloc(&b->d_kstash, (size_t)b->B * nk * b->elem);
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_kstash_i, (size_t)b->B * sizeof(int));
    if (e == hipSuccess) e = hipMemset(b->d_kstash_i, 0, (size_t)b->B * sizeof(int));      // epoch 0: never valid
    if (e != hipSuccess) { dmc_batch_destroy(b); return fail(std::string("hipMalloc kinematic stash: ") + hipGetErrorString(e), -2); }
  }
  if (b->geom.queue) {
    // queued step launches of several physics steps hand an item out in pieces (StepIO::slices; DMC_SLICES=<n>: at most n
    // pieces, 1 = whole items)
    const int nit = (b->B * b->geom.lpe + 63) / 64;
    // One queue per XCD, served by the waves that run on it (step_kernel_body): what an environment leaves in global
    // memory between its pieces then stays behind ONE L2.  An MI355X in SPX mode is 8 XCDs x 32 CUs; a device of at most
    // one XCD's worth of CUs (CPX partitions) has one L2 and one queue; anything else keeps whole items in one queue
    // per launch (DMC_XCDS overrides the count).
    b->nxcd = getenv("DMC_XCDS") ? std::max(1, std::min(8, atoi(getenv("DMC_XCDS")))) : (b->ncu == 256 ? 8 : 1);
    if (b->geom.grid < 8 * b->nxcd) b->nxcd = 1;      // (every queue needs waves of its own XCD: the dispatcher deals workgroups round)
    const bool one_l2_per_queue = b->nxcd > 1 || b->ncu <= 40 || getenv("DMC_XCDS");
    b->max_slices = getenv("DMC_SLICES") ? atoi(getenv("DMC_SLICES")) : (one_l2_per_queue ? 8 : 1);
    auto hw = [&](int n) { return b->precision == 64 ? n : (n + 1) / 2; };
    b->hand_n = hw(d.nq) + 2 * hw(d.nv) + hw(d.na) + 1;
    e = hipMalloc((void**)&b->d_prog, (size_t)nit * sizeof(int));
    if (e == hipSuccess) e = hipMemset(b->d_prog, 0, (size_t)nit * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_hand, (size_t)b->B * b->hand_n * sizeof(unsigned long long));
    if (e != hipSuccess) { dmc_batch_destroy(b); return fail(std::string("hipMalloc piece counters: ") + hipGetErrorString(e), -2); }
  }
  if (b->geom.queue && !getenv("DMC_NO_LPT")) {
    b->nitems = (b->B * b->geom.lpe + 63) / 64;
    e = hipMalloc((void**)&b->d_cost, (size_t)b->nitems * sizeof(int));
    if (e == hipSuccess) e = hipMemset(b->d_cost, 0, (size_t)b->nitems * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_order, (size_t)b->nitems * sizeof(int));
    if (e != hipSuccess) { dmc_batch_destroy(b); return fail(std::string("hipMalloc schedule: ") + hipGetErrorString(e), -2); }
    b->lpt = 1;
  }
  if (L.n_gs) {
    const size_t bytes = (size_t)b->B * L.n_gs * b->elem;
    e = hipMalloc(&b->d_gscr, bytes);
    if (e == hipSuccess) e = hipMemset(b->d_gscr, 0, bytes);
    if (e != hipSuccess) { dmc_batch_destroy(b); return fail(std::string("hipMalloc global scratch: ") + hipGetErrorString(e), -2); }
    b->tb.opts.gscr = b->d_gscr;
  }
  struct Spec { const char* name; int rows; bool is_int; };
  const int nb = d.nbody;
  const Spec specs[] = {
      {"qpos", d.nq, false}, {"qvel", d.nv, false}, {"ctrl", d.nu, false}, {"qacc_warmstart", d.nv, false},
      {"qfrc_applied", d.nv, false}, {"xfrc_applied", 6*nb, false}, {"time", 1, false}, {"act", d.na, false},
      {"mocap_pos", 3*d.nmocap, false}, {"mocap_quat", 4*d.nmocap, false},
      {"sensordata", d.nsensordata, false}, {"xpos", 3*nb, false}, {"xquat", 4*nb, false}, {"xmat", 9*nb, false},
      {"xipos", 3*nb, false}, {"geom_xpos", 3*d.ngeom, false}, {"geom_xmat", 9*d.ngeom, false},
      {"site_xpos", 3*d.nsite, false}, {"site_xmat", 9*d.nsite, false}, {"subtree_com", 3*nb, false},
      {"qacc", d.nv, false}, {"actuator_force", d.nu, false}, {"qfrc_actuator", d.nv, false},
      {"qfrc_bias", d.nv, false}, {"qfrc_constraint", d.nv, false},
      {"contact_dist", d.nconmax, false}, {"contact_pos", 3*d.nconmax, false}, {"contact_frame", 9*d.nconmax, false},
      {"contact_force", 6*d.nconmax, false}, {"cvel", 6*nb, false},
      {"ncon", 1, true}, {"nefc", 1, true}, {"solver_iter", 1, true}, {"warning", DMC_NWARNING, true},
      {"contact_geom1", d.nconmax, true}, {"contact_geom2", d.nconmax, true}, {"env_mode", 1, true}};
  for (const Spec& s : specs) {
    Field f; f.name = s.name; f.rows = s.rows; f.is_int = s.is_int; f.is_f64 = !strcmp(s.name, "time"); f.dev = nullptr; f.owned = nullptr;
    const size_t bytes = (size_t)std::max(1, s.rows) * b->B * (s.is_int ? sizeof(int) : (f.is_f64 ? sizeof(double) : b->elem));
    e = hipMalloc(&f.owned, bytes);
    if (e == hipSuccess) e = hipMemset(f.owned, 0, bytes);
    if (e != hipSuccess) { *out = nullptr; dmc_batch_destroy(b); return fail(std::string("hipMalloc field: ") + hipGetErrorString(e), -2); }
    f.dev = f.owned;
    b->index[f.name] = (int)b->fields.size();
    b->fields.push_back(f);
  }
  *out = b;
  // The stash of the position / velocity stage between legacy steps is opt-in (option "stash", or DMC_STASH=1):
  // measured on MI355X (cheetah, B = 4096) the 2 x 7.3 KB per env of stash traffic cost more than the partial
  // trailing pass it replaces (0.139 vs 0.134 ms per launch).
  if (getenv("DMC_STASH") && atoi(getenv("DMC_STASH"))) { if (dmc_batch_set_opt_int(b, "stash", 1)) { dmc_batch_destroy(b); *out = nullptr; return -2; } }
  return dmc_batch_reset(b, nullptr, -1);
}

static void prof_free(struct Profiler* p);
static void xfer_free(struct Xfer* x);
extern "C" void dmc_batch_destroy(dmc_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->prof) { prof_free(b->prof); b->prof = nullptr; }
  if (b->xfer) { xfer_free(b->xfer); b->xfer = nullptr; }
  for (Field& f : b->fields) if (f.owned) (void)hipFree(f.owned);
  if (b->d_mi) (void)hipFree(b->d_mi);
  if (b->d_mc) (void)hipFree(b->d_mc);
  if (b->d_mr) (void)hipFree(b->d_mr);
  if (b->d_layout) (void)hipFree(b->d_layout);
  if (b->d_debug) (void)hipFree(b->d_debug);
  if (b->d_debug_i) (void)hipFree(b->d_debug_i);
  if (b->d_prof) (void)hipFree(b->d_prof);
  if (b->d_stash_r) (void)hipFree(b->d_stash_r);
  if (b->d_stash_i) (void)hipFree(b->d_stash_i);
  if (b->d_epoch) (void)hipFree(b->d_epoch);
  if (b->d_eg_slot) (void)hipFree(b->d_eg_slot);
  if (b->d_ns_A) (void)hipFree(b->d_ns_A);
  if (b->d_gscr) (void)hipFree(b->d_gscr);
  if (b->d_work) (void)hipFree(b->d_work);
  if (b->d_kstash) (void)hipFree(b->d_kstash);
  if (b->d_kstash_i) (void)hipFree(b->d_kstash_i);
  if (b->d_cost) (void)hipFree(b->d_cost);
  if (b->d_order) (void)hipFree(b->d_order);
  if (b->d_prog) (void)hipFree(b->d_prog);
  if (b->d_hand) (void)hipFree(b->d_hand);
  if (b->d_task_args) (void)hipFree(b->d_task_args);
  if (b->d_trace) (void)hipFree(b->d_trace);
  if (b->d_rj_i) (void)hipFree(b->d_rj_i);
  if (b->d_rj_r) (void)hipFree(b->d_rj_r);
  delete b;
}

template <typename T>
static void fill_io(dmc_batch* b, StepIO<T>* io) {
  auto P = [&](const char* n) { return find_field(b, n)->dev; };
  io->B = b->B;
  io->qpos = (T*)P("qpos"); io->qvel = (T*)P("qvel"); io->ctrl = (T*)P("ctrl");
  io->qacc_warmstart = (T*)P("qacc_warmstart"); io->qfrc_applied = (T*)P("qfrc_applied"); io->time = (double*)P("time"); io->act = (T*)P("act"); io->prof = b->d_prof;
  io->sensordata = (T*)P("sensordata"); io->xpos = (T*)P("xpos"); io->xquat = (T*)P("xquat"); io->xmat = (T*)P("xmat");
  io->xipos = (T*)P("xipos"); io->geom_xpos = (T*)P("geom_xpos"); io->geom_xmat = (T*)P("geom_xmat");
  io->site_xpos = (T*)P("site_xpos"); io->site_xmat = (T*)P("site_xmat"); io->subtree_com = (T*)P("subtree_com");
  io->qacc = (T*)P("qacc"); io->actuator_force = (T*)P("actuator_force"); io->qfrc_actuator = (T*)P("qfrc_actuator");
  io->qfrc_bias = (T*)P("qfrc_bias"); io->qfrc_constraint = (T*)P("qfrc_constraint");
  io->contact_dist = (T*)P("contact_dist"); io->contact_pos = (T*)P("contact_pos"); io->contact_frame = (T*)P("contact_frame");
  io->contact_force = (T*)P("contact_force"); io->cvel = (T*)P("cvel");
  io->ncon = (int*)P("ncon"); io->nefc = (int*)P("nefc"); io->solver_iter = (int*)P("solver_iter");
  io->warning = (int*)P("warning"); io->contact_geom1 = (int*)P("contact_geom1"); io->contact_geom2 = (int*)P("contact_geom2");
  io->env_mode = (const int*)P("env_mode");
  io->work = b->geom.queue ? b->d_work : nullptr;
  io->cost = b->lpt ? b->d_cost : nullptr; io->order = b->lpt ? b->d_order : nullptr;
  io->prog = nullptr; io->slices = 0; io->hand = b->d_hand; io->hand_n = b->hand_n;      // (slices: launch_untimed, step launches of a queued batch only)
  io->nxcd = b->geom.queue ? b->nxcd : 1;
  io->task_args = (b->task_on && b->spec_launch) ? b->d_task_args : nullptr;
  io->trace = b->d_trace; io->trace_slot = b->d_trace ? b->trace_launch++ : 0;
  io->debug = (T*)b->d_debug; io->debug_i = b->d_debug_i; io->ndebug = b->ndebug;
  io->kstash = (T*)b->d_kstash; io->kstash_i = b->d_kstash_i;
  io->probe = (T*)b->d_probe; io->probe_geom = b->probe_geom; io->probe_cap = b->probe_cap;
  io->stash_r = b->stash_on ? (T*)b->d_stash_r : nullptr; io->stash_i = b->stash_on ? b->d_stash_i : nullptr; io->epoch = b->d_epoch;
}

// order[k] = the item handed out k-th: items by DESCENDING cost of the previous launch (1024 linear buckets of the
// largest cost; the order inside a bucket does not matter).  One workgroup; a few microseconds for 4096 items.
__global__ void __launch_bounds__(1024) order_kernel(const int* __restrict__ cost, int* __restrict__ order, int n) {
  __shared__ int hist[1024];
  __shared__ int mx;
  const int tid = threadIdx.x;
  hist[tid] = 0;
  if (tid == 0) mx = 0;
  __syncthreads();
  int m = 0;
  for (int i = tid; i < n; i += 1024) m = max(m, cost[i]);
  atomicMax(&mx, m);
  __syncthreads();
  const int top = mx;
  if (top <= 0) { for (int i = tid; i < n; i += 1024) order[i] = i; return; }      // first launch: nothing measured yet
  const float sc = 1023.0f / (float)top;
  for (int i = tid; i < n; i += 1024) atomicAdd(&hist[1023 - min(1023, (int)((float)cost[i] * sc))], 1);   // bucket 0 = costliest
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int k = 0; k < 1024; k++) { const int c = hist[k]; hist[k] = acc; acc += c; } }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) order[atomicAdd(&hist[1023 - min(1023, (int)((float)cost[i] * sc))], 1)] = i;
}

struct SeqArgs { const void* ctrl; void* qpos; void* qvel; void* sensor; int nsub; };
// ---- profiling contract (mujoco/engine.py:135-137 enable_profiling -> wrapper.enable_timer; mjData.timer[]) ------
// MuJoCo brackets mj_step / mj_forward with the mjcb_time callback and accumulates duration and call count per timer.
// Here a launch IS the step: with profiling enabled every launch is bracketed by two hipEvents on its stream; the pairs
// are resolved lazily (completed ones whenever a new launch is issued, all of them when the timer is read), so an
// asynchronous caller is not serialised.  Launches recorded into a HIP graph are not timed (events cannot be queried
// across replays).
struct TimerPair { hipEvent_t e0, e1; int which, count; };
struct Profiler {
  bool on = false;
  double duration[2] = {0, 0};      // seconds: [mjTIMER_STEP, mjTIMER_FORWARD]
  long long number[2] = {0, 0};
  std::vector<TimerPair> pending, spare;
};
static void prof_drain(Profiler* p, bool all) {
  size_t k = 0;
  for (; k < p->pending.size(); k++) {
    TimerPair& t = p->pending[k];
    if (all || p->pending.size() - k > 256) { if (hipEventSynchronize(t.e1) != hipSuccess) break; }
    else if (hipEventQuery(t.e1) != hipSuccess) break;
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) { p->duration[t.which] += 1e-3 * ms; p->number[t.which] += t.count; }
    p->spare.push_back(t);
  }
  (void)hipGetLastError();      // (hipErrorNotReady from the query is not an error of the caller)
  p->pending.erase(p->pending.begin(), p->pending.begin() + k);
}
static void prof_free(Profiler* p) {
  (void)hipDeviceSynchronize();
  for (auto* v : {&p->pending, &p->spare}) for (TimerPair& t : *v) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
  delete p;
}
static bool prof_begin(dmc_batch* b, hipStream_t stream, TimerPair* t) {
  Profiler* p = b->prof;
  if (!p || !p->on) return false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return false;
  prof_drain(p, false);
  if (!p->spare.empty()) { *t = p->spare.back(); p->spare.pop_back(); }
  else if (hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess) return false;
  return hipEventRecord(t->e0, stream) == hipSuccess;
}
static void prof_end(dmc_batch* b, hipStream_t stream, TimerPair t, int which, int count) {
  t.which = which; t.count = count;
  if (hipEventRecord(t.e1, stream) == hipSuccess) b->prof->pending.push_back(t); else b->prof->spare.push_back(t);
}
static int launch_untimed(dmc_batch* b, int nstep, int legacy, int mode, void* stream, const SeqArgs* sq);
static int launch(dmc_batch* b, int nstep, int legacy, int mode, void* stream, const SeqArgs* sq = nullptr) {
  HIP_TRY(hipSetDevice(b->device));
  TimerPair t;
  const bool timed = prof_begin(b, (hipStream_t)stream, &t);
  const int rc = launch_untimed(b, nstep, legacy, mode, stream, sq);
  // mjTIMER_STEP counts mj_step calls: a launch of mode 0 / 3 runs nstep (x n_sub_steps) of them; mj_step1 / mj_step2
  // launches count as one step per pair (on the mj_step2 half); mj_forward launches go to mjTIMER_FORWARD
  if (timed) prof_end(b, (hipStream_t)stream, t, (mode == 1 || mode == 2) ? 1 : 0,
                      mode == 0 ? nstep : mode == 3 ? nstep * (sq ? sq->nsub : 1) : mode == 5 ? 1 : mode == 4 ? 0 : 1);
  return rc;
}
static int launch_untimed(dmc_batch* b, int nstep, int legacy, int mode, void* stream, const SeqArgs* sq) {
  HIP_TRY(hipSetDevice(b->device));
  if (b->tb.opts.eg_n) b->tb.opts.eg_data = find_field(b, "env_geom")->dev;      // follows dmc_batch_bind
  b->tb.opts.xfrc = b->xfrc_on ? find_field(b, "xfrc_applied")->dev : nullptr; b->tb.opts.xfrc_B = b->B;
  if (b->tb.L.d.nmocap) { b->tb.opts.mocap_pos = find_field(b, "mocap_pos")->dev; b->tb.opts.mocap_quat = find_field(b, "mocap_quat")->dev; b->tb.opts.mocap_B = b->B; }      // follow dmc_batch_bind
  hipError_t e;
  const int nsub = sq ? sq->nsub : 1;
  // pieces of a queued Physics.step(nstep) launch (never with the full stash, whose trailing stage belongs to the last step)
  // (models of more than 16 dofs: the kernels of the small ones are built without the hand-off code, step_core.h kSlices)
  const int slices = (b->geom.queue && b->d_prog && mode == 0 && !b->stash_on && b->tb.L.d.nv > 16) ? std::min(nstep, b->max_slices) : 1;
  // longest-first hand-out pays for whole items only: with pieces a launch ends within one piece of the last claim whatever
  // the order (round 6, one box: config 3 +1.4 % without the 16 us ordering kernel in front of every launch, config 4 -0.2 %)
  if (b->lpt && slices <= 1) hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const int*)b->d_cost, b->d_order, b->nitems);
  // a specialisation plugin takes the launch unless it was built lean and the launch needs an optional feature
  const bool need_feat = legacy == 2 || b->d_probe != nullptr || b->tb.opts.integrator == DMC_INT_IMPLICITFAST;
  const bool spec = b->spec_launch && (b->spec_features || !need_feat);
  if (b->precision == 64) {
    StepIO<double> io; fill_io(b, &io);
    if (slices > 1) { io.prog = b->d_prog; io.slices = slices; io.cost = nullptr; io.order = nullptr; }
    io.ctrl_seq = sq ? (const double*)sq->ctrl : nullptr; io.qpos_seq = sq ? (double*)sq->qpos : nullptr;
    io.qvel_seq = sq ? (double*)sq->qvel : nullptr; io.sensor_seq = sq ? (double*)sq->sensor : nullptr;
    if (spec) {
      typedef int (*fn_t)(const LaunchGeom*, void*, const StepLayout*, const StepOpts<double>*, const int*, const double*, const int*, const StepIO<double>*, int, int, int, int, int);
      e = (hipError_t)((fn_t)b->spec_launch)(&b->geom, stream, b->d_layout, &b->tb.opts, b->d_mi, (const double*)b->d_mr, b->d_mc, &io, nstep, legacy, mode, b->outmask, nsub);
    } else
    e = launch_step_f64(b->geom, (hipStream_t)stream, b->d_layout, b->tb.opts, b->d_mi, (const double*)b->d_mr, b->d_mc, io, nstep, legacy, mode, b->outmask, nsub);
  } else {
    StepIO<float> io; fill_io(b, &io);
    if (slices > 1) { io.prog = b->d_prog; io.slices = slices; io.cost = nullptr; io.order = nullptr; }
    io.ctrl_seq = sq ? (const float*)sq->ctrl : nullptr; io.qpos_seq = sq ? (float*)sq->qpos : nullptr;
    io.qvel_seq = sq ? (float*)sq->qvel : nullptr; io.sensor_seq = sq ? (float*)sq->sensor : nullptr;
    if (spec) {
      typedef int (*fn_t)(const LaunchGeom*, void*, const StepLayout*, const StepOpts<float>*, const int*, const float*, const int*, const StepIO<float>*, int, int, int, int, int);
      const StepOpts<float> of = step_opts_cast<float>(b->tb.opts);
      e = (hipError_t)((fn_t)b->spec_launch)(&b->geom, stream, b->d_layout, &of, b->d_mi, (const float*)b->d_mr, b->d_mc, &io, nstep, legacy, mode, b->outmask, nsub);
    } else
    e = launch_step_f32(b->geom, (hipStream_t)stream, b->d_layout, step_opts_cast<float>(b->tb.opts), b->d_mi, (const float*)b->d_mr, b->d_mc, io, nstep, legacy, mode, b->outmask, nsub);
  }
  if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e), -2);
  return 0;
}

extern "C" int dmc_batch_attach_specialised(dmc_batch* b, const char* so_path) {
  if (!b || !so_path) return fail("null argument");
  void* h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail(std::string("cannot load ") + so_path + ": " + dlerror());
  typedef const StepLayout* (*layout_t)();
  typedef void (*info_t)(int*);
  layout_t fl = (layout_t)dlsym(h, "dmc_spec_layout");
  info_t fi = (info_t)dlsym(h, "dmc_spec_info");
  void* launch = dlsym(h, "dmc_spec_launch");
  if (!fl || !fi || !launch) { dlclose(h); return fail("not a specialisation plugin (dmc_spec_* symbols missing)"); }
  int info[8]; fi(info);
  if (info[4] != b->precision) { dlclose(h); return fail("the plugin was built for another precision"); }
  const int so = b->precision == 64 ? (int)sizeof(StepOpts<double>) : (int)sizeof(StepOpts<float>);
  const int sio = b->precision == 64 ? (int)sizeof(StepIO<double>) : (int)sizeof(StepIO<float>);
  if (info[0] != (int)sizeof(StepLayout) || info[1] != so || info[2] != sio || info[3] != DMC_MODEL_VERSION || info[7] != (int)sizeof(LaunchGeom)) {
    dlclose(h); return fail("the plugin was built from other sources than this library (struct sizes / model version differ)");
  }
  if (info[5] != b->geom.lpe) { dlclose(h); return fail("the plugin was built for another lanes-per-environment shape"); }
  if (memcmp(fl(), &b->tb.L, sizeof(StepLayout)) != 0) { dlclose(h); return fail("the plugin's layout is not this batch's (another model or other caps)"); }
  if (b->spec_so) dlclose(b->spec_so);
  b->spec_so = h; b->spec_launch = launch; b->spec_features = info[6];
  typedef int (*tb_t)();
  tb_t tb = (tb_t)dlsym(h, "dmc_spec_task_args_bytes");
  b->spec_task_bytes = tb ? tb() : 0;
  b->task_on = 0;
  return 0;
}

extern "C" int dmc_batch_set_task_args(dmc_batch* b, const void* args, int nbytes) {
  if (!b || !args) return fail("null argument");
  if (!b->spec_task_bytes) return fail("the batch's kernel carries no task epilogue (attach a plugin built with a task header)");
  if (nbytes != b->spec_task_bytes) return fail("task argument block: size differs from the one the kernel was generated with");
  HIP_TRY(hipSetDevice(b->device));
  if (!b->d_task_args) HIP_TRY(hipMalloc(&b->d_task_args, (size_t)nbytes));
  HIP_TRY(hipMemcpy(b->d_task_args, args, (size_t)nbytes, hipMemcpyHostToDevice));
  return 0;
}
extern "C" int dmc_batch_enable_task(dmc_batch* b, int on) {
  if (!b) return fail("null batch");
  if (on && (!b->spec_task_bytes || !b->d_task_args)) return fail("no task epilogue to enable (dmc_batch_set_task_args first)");
  b->task_on = on ? 1 : 0;
  return 0;
}

extern "C" int dmc_batch_step(dmc_batch* b, int nstep, int legacy_step, void* hip_stream) {
  if (!b) return fail("null batch");
  if (nstep < 1) return fail("nstep must be >= 1");
  if (legacy_step == 2 && b->tb.opts.integrator == DMC_INT_RK4) return fail("legacy_step 2 (step + mj_forward) is not implemented for RK4 models");
  return launch(b, nstep, legacy_step == 2 ? 2 : (legacy_step ? 1 : 0), 0, hip_stream);
}
extern "C" int dmc_batch_set_step_probe(dmc_batch* b, int geom_id, void* out_dev, int capacity) {
  if (!b) return fail("null batch");
  if (!out_dev || capacity < 1) { b->d_probe = nullptr; b->probe_cap = 0; return 0; }
  if (geom_id < 0 || geom_id >= b->tb.L.d.ngeom) return fail("geom id out of range");
  b->d_probe = out_dev; b->probe_geom = geom_id; b->probe_cap = capacity;
  return 0;
}
struct Field;
static int set_real(dmc_batch* b, Field* f, const double* src);
// ---- per-environment model deltas: world-fixed geoms with per-env pose / size ---------------------------------
// The reference randomises scenery per episode by editing the MJCF and recompiling (soccer RandomizedPitch,
// locomotion/soccer/pitch.py:612-690 through composer's initialize_episode_mjcf); a batch shares one compiled model,
// so the geoms that differ between environments get their pose and size from a per-env array instead of the tables.
static double geom_rbound_of(int type, const double* size) {
  switch (type) {
    case DMC_GEOM_SPHERE: return size[0];
    case DMC_GEOM_CAPSULE: return size[0] + size[1];
    case DMC_GEOM_CYLINDER: return sqrt(size[0]*size[0] + size[1]*size[1]);
    case DMC_GEOM_ELLIPSOID: return std::max(size[0], std::max(size[1], size[2]));
    case DMC_GEOM_BOX: return sqrt(size[0]*size[0] + size[1]*size[1] + size[2]*size[2]);
    default: return 0;
  }
}
// The stash epoch (StepIO::epoch) is a device int: every edit that can change what the stashed stages depend on bumps
// it with a one-thread kernel.  Host-synchronous entry points bump on the null stream and wait; dmc_batch_invalidate_async
// bumps on the caller's stream, which makes the invalidation capturable into a HIP graph together with the edit.
__global__ void epoch_bump_kernel(int* epoch) { *epoch = (*epoch == 0x7ffffff0) ? 1 : *epoch + 1; }
static int bump_epoch(dmc_batch* b, hipStream_t stream, bool wait) {
  HIP_TRY(hipSetDevice(b->device));
  hipLaunchKernelGGL(epoch_bump_kernel, dim3(1), dim3(1), 0, stream, b->d_epoch);
  HIP_TRY(hipGetLastError());
  if (wait) HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}
extern "C" int dmc_batch_set_env_geoms(dmc_batch* b, int n, const int* geom_ids) {
  if (!b || n < 1 || !geom_ids) return fail("null argument");
  if (find_field(b, "env_geom")) return fail("per-environment geoms were already declared for this batch");
  const HostModel& m = b->model->hm;
  std::vector<int> slot(m.ngeom, -1);
  for (int k = 0; k < n; k++) {
    const int g = geom_ids[k];
    if (g < 0 || g >= m.ngeom) return fail("geom id out of range");
    // world-fixed: on the worldbody, or on a jointless child of it whose frame is the world frame (PyMJCF attaches a
    // static entity -- a goal -- as such a body; the per-environment row holds the geom's WORLD pose either way)
    const int gb = m.geom_bodyid[g];
    bool fixed = gb == 0;
    if (!fixed && m.body_weldid[gb] == 0 && m.body_parentid[gb] == 0) {
      const double* bp = &m.body_pos[3*gb]; const double* bq = &m.body_quat[4*gb];
      fixed = bp[0] == 0 && bp[1] == 0 && bp[2] == 0 && bq[0] == 1 && bq[1] == 0 && bq[2] == 0 && bq[3] == 0;
    }
    if (!fixed) return fail("only world-fixed geoms (on the worldbody or on a jointless body at the world frame) can differ between environments");
    if (slot[g] >= 0) return fail("geom listed twice");
    slot[g] = k;
  }
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  Field f; f.name = "env_geom"; f.rows = 16*n; f.is_int = false; f.is_f64 = false; f.dev = nullptr; f.owned = nullptr;
  HIP_TRY(hipMalloc(&f.owned, (size_t)f.rows * b->B * b->elem));
  f.dev = f.owned;
  b->index[f.name] = (int)b->fields.size();
  b->fields.push_back(f);
  HIP_TRY(hipMalloc((void**)&b->d_eg_slot, sizeof(int) * m.ngeom));
  HIP_TRY(hipMemcpy(b->d_eg_slot, slot.data(), sizeof(int) * m.ngeom, hipMemcpyHostToDevice));
  // initial values: the model's own
  std::vector<double> host((size_t)b->B * f.rows);
  for (int k = 0; k < n; k++) {
    const int g = geom_ids[k];
    const double* q = &m.geom_quat[4*g];
    double mat[9];
    { const double w = q[0], x = q[1], y = q[2], z = q[3];
      mat[0] = w*w + x*x - y*y - z*z; mat[4] = w*w - x*x + y*y - z*z; mat[8] = w*w - x*x - y*y + z*z;
      mat[1] = 2*(x*y - w*z); mat[2] = 2*(x*z + w*y); mat[3] = 2*(x*y + w*z); mat[5] = 2*(y*z - w*x); mat[6] = 2*(x*z - w*y); mat[7] = 2*(y*z + w*x); }
    double row[16];
    for (int j = 0; j < 3; j++) row[j] = m.geom_pos[3*g + j];
    for (int j = 0; j < 9; j++) row[3 + j] = mat[j];
    for (int j = 0; j < 3; j++) row[12 + j] = m.geom_size[3*g + j];
    row[15] = m.geom_rbound[g];
    for (int e = 0; e < b->B; e++) for (int j = 0; j < 16; j++) host[(size_t)e * f.rows + 16*k + j] = row[j];
  }
  b->tb.opts.eg_slot = b->d_eg_slot; b->tb.opts.eg_n = n; b->tb.opts.eg_B = b->B;
  if (bump_epoch(b, 0, true)) return -2;
  return set_real(b, find_field(b, "env_geom"), host.data());
}
// rows of one geom slot from (pos, quat, size): what a caller writes into "env_geom" (host helper, no device work)
extern "C" int dmc_env_geom_pack(int geom_type, const double* pos, const double* quat, const double* size, double* out16) {
  if (!pos || !quat || !size || !out16) return fail("null argument");
  const double n = sqrt(quat[0]*quat[0] + quat[1]*quat[1] + quat[2]*quat[2] + quat[3]*quat[3]);
  if (!(n > 0)) return fail("zero quaternion");
  const double w = quat[0]/n, x = quat[1]/n, y = quat[2]/n, z = quat[3]/n;
  for (int j = 0; j < 3; j++) out16[j] = pos[j];
  double* mat = out16 + 3;
  mat[0] = w*w + x*x - y*y - z*z; mat[4] = w*w - x*x + y*y - z*z; mat[8] = w*w - x*x - y*y + z*z;
  mat[1] = 2*(x*y - w*z); mat[2] = 2*(x*z + w*y); mat[3] = 2*(x*y + w*z); mat[5] = 2*(y*z - w*x); mat[6] = 2*(x*z - w*y); mat[7] = 2*(y*z + w*x);
  for (int j = 0; j < 3; j++) out16[12 + j] = size[j];
  out16[15] = geom_rbound_of(geom_type, size);
  return 0;
}

// mj_step1 / mj_step2 as separate launches: the stage mj_step1 computes travels to mj_step2 through the per-env
// stash in HBM (allocated on first use)
static int ensure_stash(dmc_batch* b) {
  if (b->stash_on) return 0;
  if (dmc_batch_set_opt_int(b, "stash", 1)) return -2;
  b->stash_auto = 1;      // switched on by a step1 / step2 call, not by the caller's option: the step2 that consumes it switches it off again
  return 0;
}
extern "C" int dmc_batch_step1(dmc_batch* b, void* hip_stream) {
  if (!b) return fail("null batch");
  if (ensure_stash(b)) return -2;
  return launch(b, 1, 0, 4, hip_stream);
}
extern "C" int dmc_batch_step2(dmc_batch* b, void* hip_stream) {
  if (!b) return fail("null batch");
  if (b->tb.opts.integrator == DMC_INT_RK4) return fail("mj_step2 integrates with Euler only; RK4 models step through dmc_batch_step");
  if (ensure_stash(b)) return -2;
  const int rc = launch(b, 1, 0, 5, hip_stream);
  // a stash this pair switched on is not left on for every later launch (2 x n_keep reals per env of traffic each)
  if (b->stash_auto) { b->stash_on = 0; b->stash_auto = 0; }
  return rc;
}
extern "C" int dmc_batch_forward(dmc_batch* b, int disable_actuation, void* hip_stream) {
  if (!b) return fail("null batch");
  return launch(b, 0, 0, disable_actuation ? 2 : 1, hip_stream);
}
extern "C" int dmc_batch_rollout(dmc_batch* b, int nsteps, int n_sub_steps, const void* ctrl_seq, void* qpos_seq,
                                 void* qvel_seq, void* sensordata_seq, void* hip_stream) {
  if (!b) return fail("null batch");
  if (nsteps < 1 || n_sub_steps < 1) return fail("nsteps and n_sub_steps must be >= 1");
  SeqArgs sq = {ctrl_seq, qpos_seq, qvel_seq, sensordata_seq, n_sub_steps};
  return launch(b, nsteps, 1, 3, hip_stream, &sq);
}
// ---- observation gather table (composer/observation/updater.py:285-295, observable/mjcf.py:43) ---------------
// An MJCFFeature observable is a named slice of an mjData field; a task's enabled observables resolve once into a
// table of (field, row, corruptor) triples, and every control step ONE launch gathers them out of the SoA field
// arrays into the (B, nobs) env-major observation matrix a policy consumes.  Reads are coalesced along the env
// index (SoA rows), writes along the observation index (a 64 x 64 tile transposed through LDS).
enum { GOP_NONE = 0, GOP_GREATER = 1, GOP_TANH2 = 2, GOP_LOG1P = 3, GOP_ASINH = 4 };
struct GatherRow { int field, row, op; float prm; };
struct GatherPtrs { const void* p[16]; };
struct dmc_gather {
  dmc_batch* batch;
  int nrows;
  std::vector<std::string> field_names;   // distinct fields, slot order
  GatherRow* d_rows;
};
template <typename T>
__global__ void __launch_bounds__(256) gather_kernel(GatherPtrs ptrs, const GatherRow* __restrict__ rows, int nrows, int B, T* __restrict__ out) {
  __shared__ T tile[64][65];
  const int tx = threadIdx.x, ty = threadIdx.y;      // (64, 4)
  const int env0 = blockIdx.x * 64;
  // one 64 x 64 tile per workgroup: blockIdx.y walks the observation rows (a small batch -- soccer, B = 256: four
  // workgroups -- used to walk all its row tiles inside each of them: 61 us per control step)
  for (int k0 = blockIdx.y * 64; k0 < nrows; k0 += 64 * gridDim.y) {
    for (int kk = ty; kk < 64; kk += 4) {
      const int k = k0 + kk, env = env0 + tx;
      T v = 0;
      if (k < nrows && env < B) {
        const GatherRow r = rows[k];
        v = ((const T*)ptrs.p[r.field])[(size_t)r.row * B + env];
        if (r.op == GOP_GREATER) v = v > (T)r.prm ? (T)1 : (T)0;
        else if (r.op == GOP_TANH2) v = (T)tanh((double)(2 * v / (T)r.prm));
        else if (r.op == GOP_LOG1P) v = (T)log1p((double)v);
        else if (r.op == GOP_ASINH) v = (T)asinh((double)v);
      }
      tile[kk][tx] = v;
    }
    __syncthreads();
    for (int ee = ty; ee < 64; ee += 4) {
      const int env = env0 + ee, k = k0 + tx;
      if (env < B && k < nrows) out[(size_t)env * nrows + k] = tile[tx][ee];
    }
    __syncthreads();
  }
}
extern "C" int dmc_gather_create(dmc_batch* b, int nrows, const char* const* field_names, const int* rows, const int* ops,
                                 const double* params, dmc_gather** out) {
  if (!b || !out || nrows < 1 || !field_names || !rows) return fail("null argument");
  dmc_gather* g = new dmc_gather();
  g->batch = b; g->nrows = nrows; g->d_rows = nullptr;
  std::vector<GatherRow> h(nrows);
  for (int k = 0; k < nrows; k++) {
    Field* f = field_names[k] ? find_field(b, field_names[k]) : nullptr;
    if (!f) { delete g; return fail(std::string("unknown field: ") + (field_names[k] ? field_names[k] : "(null)")); }
    if (f->is_int || f->is_f64) { delete g; return fail(std::string("field cannot be gathered (not in batch precision): ") + f->name); }
    if (rows[k] < 0 || rows[k] >= f->rows) { delete g; return fail(std::string("row out of range for field ") + f->name); }
    int slot = -1;
    for (size_t q = 0; q < g->field_names.size(); q++) if (g->field_names[q] == f->name) slot = (int)q;
    if (slot < 0) { slot = (int)g->field_names.size(); g->field_names.push_back(f->name); }
    if (slot >= 16) { delete g; return fail("at most 16 distinct fields per gather table"); }
    const int op = ops ? ops[k] : GOP_NONE;
    if (op < GOP_NONE || op > GOP_ASINH) { delete g; return fail("unknown gather op"); }
    h[k].field = slot; h[k].row = rows[k]; h[k].op = op; h[k].prm = params ? (float)params[k] : 0.f;
  }
  hipError_t e = hipSetDevice(b->device);
  if (e == hipSuccess) e = hipMalloc((void**)&g->d_rows, sizeof(GatherRow) * nrows);
  if (e == hipSuccess) e = hipMemcpy(g->d_rows, h.data(), sizeof(GatherRow) * nrows, hipMemcpyHostToDevice);
  if (e != hipSuccess) { if (g->d_rows) (void)hipFree(g->d_rows); delete g; return fail(hipGetErrorString(e), -2); }
  *out = g;
  return 0;
}
extern "C" void dmc_gather_destroy(dmc_gather* g) {
  if (!g) return;
  if (g->d_rows) (void)hipFree(g->d_rows);
  delete g;
}
extern "C" int dmc_gather_run(dmc_gather* g, void* out, void* hip_stream) {
  if (!g || !out) return fail("null argument");
  dmc_batch* b = g->batch;
  GatherPtrs ptrs;
  for (int q = 0; q < 16; q++) ptrs.p[q] = nullptr;
  for (size_t q = 0; q < g->field_names.size(); q++) ptrs.p[q] = find_field(b, g->field_names[q].c_str())->dev;   // honours rebinding
  HIP_TRY(hipSetDevice(b->device));
  // (row tiles across blockIdx.y while the batch alone does not fill the chip: at least ~2 workgroups per CU in flight)
  const int gx = (b->B + 63) / 64, ktiles = (g->nrows + 63) / 64;
  const int gy = std::max(1, std::min(ktiles, (2 * b->ncu + gx - 1) / gx));
  const dim3 grid(gx, gy), block(64, 4);
  if (b->precision == 64) hipLaunchKernelGGL(gather_kernel<double>, grid, block, 0, (hipStream_t)hip_stream, ptrs, g->d_rows, g->nrows, b->B, (double*)out);
  else hipLaunchKernelGGL(gather_kernel<float>, grid, block, 0, (hipStream_t)hip_stream, ptrs, g->d_rows, g->nrows, b->B, (float*)out);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int dmc_batch_sync(dmc_batch* b) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

// ---- host <-> device field transfer (env-major host, SoA device) ------------------
extern "C" int dmc_batch_set_async(dmc_batch* b, const char* name, const void* src, int host_bits, void* hip_stream);
extern "C" int dmc_batch_get_async(dmc_batch* b, int n, const char* const* names, void* hip_stream);
extern "C" int dmc_batch_get_wait(dmc_batch* b, int n, void* const* dsts, int host_bits);
static bool get_in_flight(const dmc_batch* b);
static int get_real(dmc_batch* b, Field* f, double* dst) {
  const size_t n = (size_t)f->rows * b->B;
  if (!n) return 0;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  if (!get_in_flight(b)) {      // the pinned / device-transposed path, completed before returning
    const char* nm = f->name.c_str();
    if (dmc_batch_get_async(b, 1, &nm, nullptr)) return -2;
    void* d = dst;
    return dmc_batch_get_wait(b, 1, &d, 64);
  }
  if (b->precision == 64 || f->is_f64) {
    std::vector<double> tmp(n);
    HIP_TRY(hipMemcpy(tmp.data(), f->dev, n * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < f->rows; k++) for (int e = 0; e < b->B; e++) dst[(size_t)e * f->rows + k] = tmp[(size_t)k * b->B + e];
  } else {
    std::vector<float> tmp(n);
    HIP_TRY(hipMemcpy(tmp.data(), f->dev, n * sizeof(float), hipMemcpyDeviceToHost));
    for (int k = 0; k < f->rows; k++) for (int e = 0; e < b->B; e++) dst[(size_t)e * f->rows + k] = tmp[(size_t)k * b->B + e];
  }
  return 0;
}
static int set_real(dmc_batch* b, Field* f, const double* src) {
  const size_t n = (size_t)f->rows * b->B;
  if (!n) return 0;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  // (ctrl / qfrc_applied / xfrc_applied are inputs of the acceleration stage only; every other field bumps the stash epoch)
  // the synchronous form of dmc_batch_set_async: pinned staging, device-side transposition, done before returning
  if (dmc_batch_set_async(b, f->name.c_str(), src, 64, nullptr)) return -2;
  HIP_TRY(hipStreamSynchronize(nullptr));
  return 0;
}
// ---- asynchronous host transfers -------------------------------------------------------------------------------
// The host side of the boundary is env-major ((B, rows): what numpy callers hold), the device fields are SoA
// ((rows, B)).  dmc_batch_set / dmc_batch_get transpose and convert element by element on one host thread around
// synchronous pageable copies.  Here: the host only converts contiguously into PINNED staging (or not at all: fp32 on
// the wire for fp32 batches), copies are hipMemcpyAsync on the caller's stream, the transposition is a device kernel
// (LDS tile, coalesced on both sides), and a get of several fields is ONE device-to-host copy and one wait.
struct Xfer {
  static constexpr int kSlots = 4;
  void* h_in[kSlots] = {}; void* d_in[kSlots] = {}; size_t cap_in[kSlots] = {}; hipEvent_t ev_in[kSlots] = {}; int next = 0;
  void* h_out = nullptr; void* d_out = nullptr; size_t cap_out = 0; hipEvent_t ev_out = nullptr;
  std::vector<std::pair<Field*, size_t>> pending;      // fields of the enqueued get and their offsets (elements) in the staging
  bool out_in_flight = false;
};
static void xfer_free(Xfer* x) {
  (void)hipDeviceSynchronize();
  for (int k = 0; k < Xfer::kSlots; k++) { if (x->h_in[k]) (void)hipHostFree(x->h_in[k]); if (x->d_in[k]) (void)hipFree(x->d_in[k]); if (x->ev_in[k]) (void)hipEventDestroy(x->ev_in[k]); }
  if (x->h_out) (void)hipHostFree(x->h_out);
  if (x->d_out) (void)hipFree(x->d_out);
  if (x->ev_out) (void)hipEventDestroy(x->ev_out);
  delete x;
}
// out[c * ldo + r] = in[r * ldi + c] for r < nr, c < nc  (32 x 32 tiles through LDS).  `in` / `out` may be PINNED HOST
// memory (device-mapped): the set kernel reads the caller's staged (B, rows) array over PCIe and the get kernel writes
// the (B, rows) arrays straight into the host staging -- no separate hipMemcpyAsync on either side.
template <typename T>
__device__ inline void transpose_tile(const T* __restrict__ in, T* __restrict__ out, int nr, int nc, int ldi, int ldo, int bx, int by, T (*tile)[33]) {
  const int c0 = bx * 32, r0 = by * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) { const int r = r0 + k, c = c0 + tx; if (r < nr && c < nc) tile[k][tx] = in[(size_t)r * ldi + c]; }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) { const int c = c0 + k, r = r0 + tx; if (r < nr && c < nc) out[(size_t)c * ldo + r] = tile[tx][k]; }
}
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int nr, int nc, int ldi, int ldo) {
  __shared__ T tile[32][33];
  transpose_tile<T>(in, out, nr, nc, ldi, ldo, blockIdx.x, blockIdx.y, tile);
}
template <typename T>
static void launch_transpose(const void* in, void* out, int nr, int nc, int ldi, int ldo, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel<T>, dim3((nc + 31) / 32, (nr + 31) / 32), dim3(256), 0, s, (const T*)in, (T*)out, nr, nc, ldi, ldo);
}
// every field of a get in ONE launch: blockIdx.z = field
struct GetTable { const void* src[8]; void* dst[8]; int rows[8]; int wide[8]; int n; };
__global__ void __launch_bounds__(256) get_pack_kernel(GetTable t, int B) {
  __shared__ double tile64[32][33];
  const int f = blockIdx.z;
  if ((int)blockIdx.y * 32 >= t.rows[f]) return;
  if (t.wide[f]) transpose_tile<double>((const double*)t.src[f], (double*)t.dst[f], t.rows[f], B, B, t.rows[f], blockIdx.x, blockIdx.y, tile64);
  else transpose_tile<float>((const float*)t.src[f], (float*)t.dst[f], t.rows[f], B, B, t.rows[f], blockIdx.x, blockIdx.y, (float (*)[33])tile64);
}
static bool get_in_flight(const dmc_batch* b) { return b->xfer && b->xfer->out_in_flight; }
static size_t field_elem(const dmc_batch* b, const Field* f) { return f->is_int ? sizeof(int32_t) : (b->precision == 64 || f->is_f64) ? sizeof(double) : sizeof(float); }
extern "C" int dmc_batch_set_async(dmc_batch* b, const char* name, const void* src, int host_bits, void* hip_stream) {
  if (!b || !name || !src) return fail("null argument");
  if (host_bits != 64 && host_bits != 32) return fail("host_bits must be 64 or 32");
  Field* f = find_field(b, name);
  if (!f || f->is_int) return fail(std::string("unknown real field: ") + name);
  const size_t n = (size_t)f->rows * b->B;
  if (!n) return 0;
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(hipSetDevice(b->device));
  if (!b->xfer) b->xfer = new Xfer();
  Xfer* x = b->xfer;
  const int k = x->next; x->next = (k + 1) % Xfer::kSlots;
  const size_t es = field_elem(b, f), bytes = n * es;
  if (x->cap_in[k] < bytes) {
    if (x->ev_in[k]) HIP_TRY(hipEventSynchronize(x->ev_in[k]));
    if (x->h_in[k]) (void)hipHostFree(x->h_in[k]);
    if (x->d_in[k]) (void)hipFree(x->d_in[k]);
    x->h_in[k] = nullptr; x->d_in[k] = nullptr; x->cap_in[k] = 0;
    HIP_TRY(hipHostMalloc(&x->h_in[k], bytes, hipHostMallocMapped));
    x->cap_in[k] = bytes;
    if (!x->ev_in[k]) HIP_TRY(hipEventCreateWithFlags(&x->ev_in[k], hipEventDisableTiming));
  } else HIP_TRY(hipEventSynchronize(x->ev_in[k]));      // the copy that last used this slot (four sets ago) has left the staging
  // host: contiguous conversion into the pinned slot (no transposition); the caller's array is free again on return
  if (es == 8) { double* d = (double*)x->h_in[k]; if (host_bits == 64) std::memcpy(d, src, bytes); else { const float* p = (const float*)src; for (size_t i = 0; i < n; i++) d[i] = p[i]; } }
  else { float* d = (float*)x->h_in[k]; if (host_bits == 32) std::memcpy(d, src, bytes); else { const double* p = (const double*)src; for (size_t i = 0; i < n; i++) d[i] = (float)p[i]; } }
  void* hin = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&hin, x->h_in[k], 0));
  if (es == 8) launch_transpose<double>(hin, f->dev, b->B, f->rows, f->rows, b->B, st); else launch_transpose<float>(hin, f->dev, b->B, f->rows, f->rows, b->B, st);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(x->ev_in[k], st));
  if (f->name == "xfrc_applied") b->xfrc_on = 1;
  if (f->name != "ctrl" && f->name != "qfrc_applied" && f->name != "xfrc_applied") return bump_epoch(b, st, false);      // stream-ordered with the edit
  return 0;
}
extern "C" int dmc_batch_get_async(dmc_batch* b, int n, const char* const* names, void* hip_stream) {
  if (!b || n < 1 || !names) return fail("null argument");
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(hipSetDevice(b->device));
  if (!b->xfer) b->xfer = new Xfer();
  Xfer* x = b->xfer;
  if (x->out_in_flight) return fail("a get is already enqueued: call dmc_batch_get_wait first");
  x->pending.clear();
  size_t bytes = 0;
  for (int i = 0; i < n; i++) {
    Field* f = find_field(b, names[i]);
    if (!f) return fail(std::string("unknown field: ") + (names[i] ? names[i] : "(null)"));      // (int32 fields travel as they are)
    bytes = (bytes + 7) / 8 * 8;
    x->pending.push_back({f, bytes});
    bytes += (size_t)f->rows * b->B * field_elem(b, f);
  }
  if (n > 8) return fail("at most 8 fields per get");
  if (x->cap_out < bytes) {
    if (x->h_out) (void)hipHostFree(x->h_out);
    x->h_out = nullptr; x->cap_out = 0;
    HIP_TRY(hipHostMalloc(&x->h_out, bytes, hipHostMallocMapped));
    x->cap_out = bytes;
  }
  if (!x->ev_out) HIP_TRY(hipEventCreateWithFlags(&x->ev_out, hipEventDisableTiming));
  void* hout = nullptr;
  if (bytes) HIP_TRY(hipHostGetDevicePointer(&hout, x->h_out, 0));
  GetTable t; t.n = n;
  int maxrows = 0;
  for (int i = 0; i < 8; i++) { t.src[i] = nullptr; t.dst[i] = nullptr; t.rows[i] = 0; t.wide[i] = 0; }
  for (int i = 0; i < n; i++) {
    Field* f = x->pending[i].first;
    t.src[i] = f->dev; t.dst[i] = (char*)hout + x->pending[i].second; t.rows[i] = f->rows; t.wide[i] = field_elem(b, f) == 8;
    maxrows = std::max(maxrows, f->rows);
  }
  if (maxrows) hipLaunchKernelGGL(get_pack_kernel, dim3((b->B + 31) / 32, (maxrows + 31) / 32, n), dim3(256), 0, st, t, b->B);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(x->ev_out, st));
  x->out_in_flight = true;
  return 0;
}
extern "C" int dmc_batch_get_wait(dmc_batch* b, int n, void* const* dsts, int host_bits) {
  if (!b || !dsts) return fail("null argument");
  if (host_bits != 64 && host_bits != 32) return fail("host_bits must be 64 or 32");
  Xfer* x = b->xfer;
  if (!x || !x->out_in_flight) return fail("no get is enqueued");
  if ((size_t)n != x->pending.size()) return fail("dmc_batch_get_wait: the number of destinations differs from the enqueued get");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipEventSynchronize(x->ev_out));
  x->out_in_flight = false;
  for (int i = 0; i < n; i++) {
    Field* f = x->pending[i].first;
    const size_t cnt = (size_t)f->rows * b->B, es = field_elem(b, f);
    if (!cnt || !dsts[i]) continue;
    const void* srcp = (const char*)x->h_out + x->pending[i].second;
    if (f->is_int) { std::memcpy(dsts[i], srcp, cnt * 4); continue; }      // int32 whatever host_bits says
    if (es == 8) { const double* p = (const double*)srcp; if (host_bits == 64) std::memcpy(dsts[i], p, cnt * 8); else { float* d = (float*)dsts[i]; for (size_t k = 0; k < cnt; k++) d[k] = (float)p[k]; } }
    else { const float* p = (const float*)srcp; if (host_bits == 32) std::memcpy(dsts[i], p, cnt * 4); else { double* d = (double*)dsts[i]; for (size_t k = 0; k < cnt; k++) d[k] = p[k]; } }
  }
  return 0;
}
extern "C" const void* dmc_batch_get_staged(dmc_batch* b, int i) {
  if (!b || !b->xfer || b->xfer->out_in_flight || i < 0 || (size_t)i >= b->xfer->pending.size()) { fail("no completed get holds that field"); return nullptr; }
  return (const char*)b->xfer->h_out + b->xfer->pending[i].second;
}
extern "C" int dmc_batch_field_rows(const dmc_batch* b, const char* name, int* rows, int* is_int) {
  if (!b || !name) return fail("null argument");
  Field* f = find_field(const_cast<dmc_batch*>(b), name);
  if (!f) return fail(std::string("unknown field: ") + name);
  if (rows) *rows = f->rows;
  if (is_int) *is_int = f->is_int;
  return 0;
}
extern "C" int dmc_batch_get(dmc_batch* b, const char* name, double* dst) {
  if (!b || !name || !dst) return fail("null argument");
  Field* f = find_field(b, name);
  if (!f || f->is_int) return fail(std::string("unknown real field: ") + name);
  return get_real(b, f, dst);
}
extern "C" int dmc_batch_set(dmc_batch* b, const char* name, const double* src) {
  if (!b || !name || !src) return fail("null argument");
  Field* f = find_field(b, name);
  if (!f || f->is_int) return fail(std::string("unknown real field: ") + name);
  return set_real(b, f, src);
}
extern "C" int dmc_batch_get_int(dmc_batch* b, const char* name, int32_t* dst) {
  if (!b || !name || !dst) return fail("null argument");
  Field* f = find_field(b, name);
  if (!f || !f->is_int) return fail(std::string("unknown int field: ") + name);
  const size_t n = (size_t)f->rows * b->B;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<int32_t> tmp(n);
  HIP_TRY(hipMemcpy(tmp.data(), f->dev, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int k = 0; k < f->rows; k++) for (int e = 0; e < b->B; e++) dst[(size_t)e * f->rows + k] = tmp[(size_t)k * b->B + e];
  return 0;
}
extern "C" int dmc_batch_set_int(dmc_batch* b, const char* name, const int32_t* src) {
  if (!b || !name || !src) return fail("null argument");
  Field* f = find_field(b, name);
  if (!f || !f->is_int) return fail(std::string("unknown int field: ") + name);
  const size_t n = (size_t)f->rows * b->B;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<int32_t> tmp(n);
  for (int k = 0; k < f->rows; k++) for (int e = 0; e < b->B; e++) tmp[(size_t)k * b->B + e] = src[(size_t)e * f->rows + k];
  HIP_TRY(hipMemcpy(f->dev, tmp.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
  return 0;
}
extern "C" void* dmc_batch_device_ptr(dmc_batch* b, const char* name) {
  if (!b || !name) { fail("null argument"); return nullptr; }
  Field* f = find_field(b, name);
  if (!f) { fail(std::string("unknown field: ") + name); return nullptr; }
  if (f->name == "xfrc_applied") b->xfrc_on = 1;
  return f->dev;
}
extern "C" int dmc_batch_bind(dmc_batch* b, const char* name, void* device_ptr) {
  if (!b || !name) return fail("null argument");
  Field* f = find_field(b, name);
  if (!f) return fail(std::string("unknown field: ") + name);
  f->dev = device_ptr ? device_ptr : f->owned;
  if (f->name == "xfrc_applied") b->xfrc_on = 1;
  if (f->name != "ctrl" && f->name != "qfrc_applied") { if (bump_epoch(b, 0, true)) return -2; }
  return 0;
}
extern "C" int dmc_batch_invalidate(dmc_batch* b) {
  if (!b) return fail("null batch");
  return bump_epoch(b, 0, true);
}
extern "C" int dmc_batch_invalidate_async(dmc_batch* b, void* hip_stream) {
  if (!b) return fail("null batch");
  return bump_epoch(b, (hipStream_t)hip_stream, false);
}
extern "C" int dmc_batch_set_output_mask(dmc_batch* b, int mask) {
  if (!b) return fail("null batch");
  if (bump_epoch(b, 0, true)) return -2;
  b->outmask = mask;
  return 0;
}
extern "C" int dmc_batch_set_opt_int(dmc_batch* b, const char* name, int value) {
  if (!b || !name) return fail("null argument");
  StepOpts<double>& o = b->tb.opts;
  if (bump_epoch(b, 0, true)) return -2;
  if (!strcmp(name, "stash")) {
    if (value && !b->d_stash_r) {
      const StepLayout& L = b->tb.L;
      HIP_TRY(hipSetDevice(b->device));
      const size_t nr = (size_t)std::max(1, L.n_keep) * b->B * b->elem, ni = (size_t)(L.n_si + 4) * b->B * sizeof(int);
      HIP_TRY(hipMalloc(&b->d_stash_r, nr));
      HIP_TRY(hipMalloc((void**)&b->d_stash_i, ni));
      HIP_TRY(hipMemset(b->d_stash_i, 0, ni));
    }
    b->stash_on = value ? 1 : 0;
    return 0;
  }
  if (!strcmp(name, "disableflags")) {
    if (b->tb.has_unsupported_pairs && !(value & (DMC_DSBL_CONTACT | DMC_DSBL_CONSTRAINT)))
      return fail("cannot enable contacts: the model has geom pair types the collision kernel does not implement");
    o.disableflags = value;
  }
  else if (!strcmp(name, "islands")) o.islands = value < 0 ? -1 : (value ? 1 : 0);      // per-island solves: 1 / 0 / -1 = by precision
  else if (!strcmp(name, "iterations")) o.iterations = value;
  else if (!strcmp(name, "ls_iterations")) o.ls_iterations = value;
  else if (!strcmp(name, "noslip_iterations")) {
    if (value > 0 && !b->tb.L.d.nslip) return fail("noslip needs scratch the batch was created without: compile the model with noslip_iterations > 0");
    o.noslip_iterations = value;
  }
  else return fail(std::string("unknown int option: ") + name);
  return 0;
}
extern "C" int dmc_batch_set_opt_real(dmc_batch* b, const char* name, double value) {
  if (!b || !name) return fail("null argument");
  StepOpts<double>& o = b->tb.opts;
  if (bump_epoch(b, 0, true)) return -2;
  if (!strcmp(name, "timestep")) { o.timestep = value; o.timestep_d = value; }   // timestep_d: what data.time advances by
  else if (!strcmp(name, "tolerance")) o.tolerance = value;
  else if (!strcmp(name, "ls_tolerance")) o.ls_tolerance = value;
  else if (!strcmp(name, "noslip_tolerance")) o.noslip_tolerance = value;
  else if (!strcmp(name, "gravity_x")) o.gravity[0] = value;
  else if (!strcmp(name, "gravity_y")) o.gravity[1] = value;
  else if (!strcmp(name, "gravity_z")) o.gravity[2] = value;
  else return fail(std::string("unknown real option: ") + name);
  return 0;
}

// Model constants that tasks rewrite between episodes (straight copies of mjModel arrays
// in the kernel's tables; anything the tables derive -- contact-pair mixing, inertias,
// invweight0 -- is fixed at batch creation, as mj_setConst-dependent fields are in MuJoCo).
extern "C" int dmc_batch_set_model_real(dmc_batch* b, const char* name, const double* values, int count) {
  if (!b || !name || !values) return fail("null argument");
  const StepLayout& L = b->tb.L;
  const StepDims& d = L.d;
  struct Slot { const char* name; int off, cnt, stride, ncopy; };   // stride: source elements per table element group
  const Slot slots[] = {
      {"dof_damping", L.mr_dof_damping, d.nv, 1, 1},   {"jnt_stiffness", L.mr_jnt_stiffness, d.njnt, 1, 1},
      {"jnt_range", L.mr_jnt_range, 2 * d.njnt, 1, 1}, {"jnt_margin", L.mr_jnt_margin, d.njnt, 1, 1},
      {"qpos_spring", L.mr_qpos_spring, d.nq, 1, 1},   {"site_pos", L.mr_site_pos, 3 * d.nsite, 1, 1},
      {"site_quat", L.mr_site_quat, 4 * d.nsite, 1, 1}, {"site_size", L.mr_site_size, 3 * d.nsite, 1, 1},
      {"actuator_ctrlrange", L.mr_act_ctrlrange, 2 * d.nu, 1, 1},
      {"actuator_forcerange", L.mr_act_forcerange, 2 * d.nu, 1, 1},
      {"wrap_prm", L.mr_wrap_prm, d.nwrap, 1, 1},
      {"body_pos", L.mr_body_pos, 3 * d.nbody, 1, 1}, {"body_quat", L.mr_body_quat, 4 * d.nbody, 1, 1},
      // geom frames / sizes (suite/reacher.py:88-94, suite/fish.py:150-154 move and resize their target geom); like a
      // write to mjModel, nothing derived at compile time follows (body inertias, geom_rbound, contact-pair mixing)
      {"geom_pos", L.mr_geom_pos, 3 * d.ngeom, 1, 1}, {"geom_quat", L.mr_geom_quat, 4 * d.ngeom, 1, 1},
      {"geom_size", L.mr_geom_size, 3 * d.ngeom, 1, 1},
  };
  for (const Slot& s : slots) {
    if (strcmp(name, s.name)) continue;
    if (count != s.cnt) return fail(std::string("wrong element count for model field ") + name);
    for (int i = 0; i < count; i++) b->tb.mr[s.off + i] = values[i];
    if (bump_epoch(b, 0, true)) return -2;
    if (!strcmp(name, "dof_damping")) {
      b->tb.opts.any_damping = 0;
      for (int i = 0; i < count; i++) { if (values[i] < 0) return fail("negative dof_damping"); if (values[i] > 0) b->tb.opts.any_damping = 1; }
    }
    hipError_t e = hipSetDevice(b->device);
    if (e != hipSuccess) return fail(hipGetErrorString(e));
    if (dmc_batch_sync(b)) return -2;
    return upload_tables(b);
  }
  return fail(std::string("model field cannot be changed after batch creation: ") + name);
}

extern "C" int dmc_batch_reset(dmc_batch* b, const uint8_t* env_mask, int keyframe) {
  if (!b) return fail("null batch");
  const HostModel& m = b->model->hm;
  if (keyframe >= m.nkey) return fail("keyframe out of range");
  const int B = b->B;
  auto reset_real = [&](const char* name, const double* init, int rows) -> int {
    Field* f = find_field(b, name);
    if (!rows) return 0;
    std::vector<double> host((size_t)B * rows);
    if (env_mask) { if (get_real(b, f, host.data())) return -2; }
    for (int e = 0; e < B; e++) if (!env_mask || env_mask[e]) for (int k = 0; k < rows; k++) host[(size_t)e * rows + k] = init ? init[k] : 0.0;
    return set_real(b, f, host.data());
  };
  const double* q0 = keyframe >= 0 ? &m.key_qpos[(size_t)keyframe * m.nq] : m.qpos0.data();
  const double* v0 = keyframe >= 0 ? &m.key_qvel[(size_t)keyframe * m.nv] : nullptr;
  const double* c0 = keyframe >= 0 ? &m.key_ctrl[(size_t)keyframe * m.nu] : nullptr;
  if (reset_real("qpos", q0, m.nq)) return -2;
  if (reset_real("qvel", v0, m.nv)) return -2;
  if (reset_real("ctrl", c0, m.nu)) return -2;
  if (reset_real("qacc_warmstart", nullptr, m.nv)) return -2;
  if (reset_real("qfrc_applied", nullptr, m.nv)) return -2;
  if (b->xfrc_on && reset_real("xfrc_applied", nullptr, 6*m.nbody)) return -2;
  // mj_resetDataKeyframe restores time, the activations and the mocap poses of the keyframe as well
  if (reset_real("time", keyframe >= 0 ? &m.key_time[keyframe] : nullptr, 1)) return -2;
  if (reset_real("act", keyframe >= 0 && m.na ? &m.key_act[(size_t)keyframe * m.na] : nullptr, m.na)) return -2;
  if (m.nmocap) {      // mj_resetData: the mocap poses start at the bodies' model poses
    std::vector<double> mp(3 * (size_t)m.nmocap), mq(4 * (size_t)m.nmocap);
    for (int i = 0; i < m.nbody; i++) if (m.body_mocapid[i] >= 0) {
      for (int k = 0; k < 3; k++) mp[3*m.body_mocapid[i] + k] = m.body_pos[3*i + k];
      for (int k = 0; k < 4; k++) mq[4*m.body_mocapid[i] + k] = m.body_quat[4*i + k];
    }
    if (keyframe >= 0) {
      std::copy(m.key_mpos.begin() + (size_t)keyframe * 3 * m.nmocap, m.key_mpos.begin() + (size_t)(keyframe + 1) * 3 * m.nmocap, mp.begin());
      std::copy(m.key_mquat.begin() + (size_t)keyframe * 4 * m.nmocap, m.key_mquat.begin() + (size_t)(keyframe + 1) * 4 * m.nmocap, mq.begin());
    }
    if (reset_real("mocap_pos", mp.data(), 3*m.nmocap)) return -2;
    if (reset_real("mocap_quat", mq.data(), 4*m.nmocap)) return -2;
  }
  // mj_resetData clears warnings as well
  Field* w = find_field(b, "warning");
  std::vector<int32_t> wh((size_t)B * DMC_NWARNING, 0);
  if (env_mask) { if (dmc_batch_get_int(b, "warning", wh.data())) return -2; for (int e = 0; e < B; e++) if (env_mask[e]) for (int k = 0; k < DMC_NWARNING; k++) wh[(size_t)e * DMC_NWARNING + k] = 0; }
  (void)w;
  return dmc_batch_set_int(b, "warning", wh.data());
}

extern "C" int dmc_batch_info(const dmc_batch* b, int* info) {
  if (!b || !info) return fail("null argument");
  const StepLayout& L = b->tb.L;
  info[0] = b->B; info[1] = b->precision; info[2] = b->geom.lpe; info[3] = b->geom.waves; info[4] = b->geom.envs_per_block;
  info[5] = b->geom.lds_bytes; info[6] = b->geom.grid; info[7] = L.d.nconmax; info[8] = L.d.njmax;
  info[9] = (int)((size_t)L.n_sr * b->elem + (size_t)L.n_si * sizeof(int));
  info[10] = b->spec_launch ? 1000 : b->geom.static_id;      // 1000: a specialisation plugin is attached
  info[11] = L.d.kmax;
  info[12] = b->geom.lds_bytes - b->geom.envs_per_block * info[9];
  { long blocks = (160L * 1024) / b->geom.lds_bytes; if (blocks * b->geom.waves > 8) blocks = 8 / b->geom.waves; info[13] = (int)(blocks * b->geom.envs_per_block); }
  info[14] = L.d.njdense; info[15] = L.d.njcon; info[16] = b->stash_on; info[17] = (int)((size_t)L.n_keep * b->elem + (size_t)(L.n_si + 4) * sizeof(int));
  info[18] = (int)((size_t)L.n_gs * b->elem);
  info[19] = b->geom.queue;
  return 0;
}

extern "C" int dmc_batch_enable_profiling(dmc_batch* b, int enabled) {
  if (!b) return fail("null batch");
  if (!b->prof) b->prof = new Profiler();
  b->prof->on = enabled != 0;
  return 0;
}
extern "C" int dmc_batch_get_timer(dmc_batch* b, int timer, double* duration_s, long long* number) {
  if (!b || !duration_s || !number) return fail("null argument");
  if (timer != 0 && timer != 1) return fail("timer must be 0 (mjTIMER_STEP) or 1 (mjTIMER_FORWARD)");
  *duration_s = 0; *number = 0;
  if (!b->prof) return 0;
  HIP_TRY(hipSetDevice(b->device));
  prof_drain(b->prof, true);
  *duration_s = b->prof->duration[timer]; *number = b->prof->number[timer];
  return 0;
}
extern "C" int dmc_batch_time_steps(dmc_batch* b, int nstep, int legacy_step, int reps, void* hip_stream, float* ms_per_launch) {
  if (!b || !ms_per_launch || reps < 1) return fail("bad argument");
  HIP_TRY(hipSetDevice(b->device));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, (hipStream_t)hip_stream));
  for (int r = 0; r < reps; r++) { int rc = dmc_batch_step(b, nstep, legacy_step, hip_stream); if (rc) return rc; }
  HIP_TRY(hipEventRecord(e1, (hipStream_t)hip_stream));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *ms_per_launch = ms / reps;
  return 0;
}

extern "C" int dmc_batch_debug_enable(dmc_batch* b, int n) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->device));
  if (b->d_debug) { (void)hipFree(b->d_debug); b->d_debug = nullptr; }
  if (b->d_debug_i) { (void)hipFree(b->d_debug_i); b->d_debug_i = nullptr; }
  b->ndebug = 0;
  if (n <= 0) return 0;
  if (n > b->B) n = b->B;
  const StepLayout& L = b->tb.L;
  HIP_TRY(hipMalloc(&b->d_debug, (size_t)(L.n_sr + L.n_gs) * n * b->elem));
  HIP_TRY(hipMalloc((void**)&b->d_debug_i, (size_t)L.n_si * n * sizeof(int)));
  HIP_TRY(hipMemset(b->d_debug, 0, (size_t)(L.n_sr + L.n_gs) * n * b->elem));
  HIP_TRY(hipMemset(b->d_debug_i, 0, (size_t)L.n_si * n * sizeof(int)));
  b->ndebug = n;
  return 0;
}
extern "C" int dmc_batch_debug_get(dmc_batch* b, const char* scratch_name, int env, double* dst, int* count) {
  if (!b || !scratch_name || !dst || !count) return fail("null argument");
  if (env < 0 || env >= b->ndebug) return fail("env outside the debug window");
  int off, cnt, kind;
  if (!step_layout_find(&b->tb.L, scratch_name, &off, &cnt, &kind)) return fail(std::string("unknown scratch array: ") + scratch_name);
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  const int n = b->ndebug;
  const size_t total = (size_t)cnt * n;
  if (kind) {
    std::vector<int> tmp(total);
    HIP_TRY(hipMemcpy(tmp.data(), b->d_debug_i + (size_t)off * n, total * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < cnt; i++) dst[i] = tmp[(size_t)i * n + env];
  } else if (b->precision == 64) {
    std::vector<double> tmp(total);
    HIP_TRY(hipMemcpy(tmp.data(), (double*)b->d_debug + (size_t)off * n, total * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < cnt; i++) dst[i] = tmp[(size_t)i * n + env];
  } else {
    std::vector<float> tmp(total);
    HIP_TRY(hipMemcpy(tmp.data(), (float*)b->d_debug + (size_t)off * n, total * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < cnt; i++) dst[i] = tmp[(size_t)i * n + env];
  }
  *count = cnt;
  return 0;
}

// ---- per-episode joint randomisation on the device -------------------------------------------------------------
// suite/utils/randomizers.py:35-88 (randomize_limited_and_rotational_joints) and the draws of
// suite/cheetah.py:66-69 / suite/quadruped.py:_find_non_contacting_height for a whole batch, without a host round
// trip and without a finite pool of start states: a counter-based generator (Philox4x32-10, Salmon et al. 2011) keyed
// by the 64-bit seed with counter (draw number of the env, joint, block, env) -- every (env, episode, joint) has its own
// stream whatever the batch size, launch order or mask, and different seeds give different streams for every env
// (keying by seed ^ env made seeds below B permutations of one another).
struct Philox { uint32_t c[4]; };
__host__ __device__ inline Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox o = {{c0, c1, c2, c3}};
  return o;
}
__device__ inline double philox_u01(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }      // in (0, 1)
__device__ inline void box_muller(double u0, double u1, double* n0, double* n1) {
  const double r = sqrt(-2.0 * log(u0)), a = 6.283185307179586476925286766559 * u1;
  *n0 = r * cos(a); *n1 = r * sin(a);
}
template <typename T>
__global__ void __launch_bounds__(256) randomize_joints_kernel(T* __restrict__ qpos, int B, int njnt, const int* __restrict__ ji,
                                                               const double* __restrict__ jr, uint32_t seed_lo, uint32_t seed_hi,
                                                               int* __restrict__ draw, const int* __restrict__ mask, int flags) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B || (mask && !mask[env])) return;
  const uint32_t k = (uint32_t)draw[env];
  draw[env] = (int)(k + 1);
  const uint32_t key0 = seed_lo, key1 = seed_hi, cenv = (uint32_t)env;   // key = the seed alone; env is a counter word
  for (int j = 0; j < njnt; j++) {
    const int type = ji[j], adr = ji[njnt + j], limited = ji[2*njnt + j];
    const double lo = jr[2*j], hi = jr[2*j + 1];
    const Philox x = philox4x32_10(k, (uint32_t)j, 0u, cenv, key0, key1);
    if (type == DMC_JNT_HINGE || type == DMC_JNT_SLIDE) {
      if (limited) { if (flags & DMC_RAND_LIMITED) qpos[(size_t)adr*B + env] = (T)(lo + (hi - lo) * philox_u01(x.c[0])); }
      else if (type == DMC_JNT_HINGE && (flags & DMC_RAND_UNLIMITED_HINGE))
        qpos[(size_t)adr*B + env] = (T)(-3.14159265358979323846 + 6.283185307179586476925286766559 * philox_u01(x.c[0]));
    } else if (type == DMC_JNT_BALL && limited) {
      if (!(flags & DMC_RAND_LIMITED)) continue;
      // random_limited_quaternion: axis ~ normalised N(0, I), angle ~ U(0, range max)
      const Philox y = philox4x32_10(k, (uint32_t)j, 1u, cenv, key0, key1);
      double n[4];
      box_muller(philox_u01(x.c[0]), philox_u01(x.c[1]), &n[0], &n[1]);
      box_muller(philox_u01(x.c[2]), philox_u01(x.c[3]), &n[2], &n[3]);
      const double nn = sqrt(n[0]*n[0] + n[1]*n[1] + n[2]*n[2]), ang = philox_u01(y.c[0]) * hi, sn = sin(0.5 * ang) / nn;
      qpos[(size_t)adr*B + env] = (T)cos(0.5 * ang);
      for (int c = 0; c < 3; c++) qpos[(size_t)(adr + 1 + c)*B + env] = (T)(n[c] * sn);
    } else if (type == DMC_JNT_BALL || type == DMC_JNT_FREE) {
      if (!(flags & DMC_RAND_QUATERNION)) continue;
      // ball joints: normalised N(0, I) (uniform on the 3-sphere); free joints: normalised U(0, 1)^4 as the reference
      // draws them (randomizers.py:84-88 keeps `rand` on purpose) unless DMC_RAND_FREE_NORMAL asks for the sphere
      double q[4];
      if (type == DMC_JNT_BALL || (flags & DMC_RAND_FREE_NORMAL)) {
        box_muller(philox_u01(x.c[0]), philox_u01(x.c[1]), &q[0], &q[1]);
        box_muller(philox_u01(x.c[2]), philox_u01(x.c[3]), &q[2], &q[3]);
      } else {
        for (int c = 0; c < 4; c++) q[c] = philox_u01(x.c[c]);
      }
      const double inv = 1.0 / sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
      const int a0 = adr + (type == DMC_JNT_FREE ? 3 : 0);
      for (int c = 0; c < 4; c++) qpos[(size_t)(a0 + c)*B + env] = (T)(q[c] * inv);
    }
  }
}
extern "C" int dmc_batch_randomize_joints(dmc_batch* b, uint64_t seed, int32_t* d_draw, const int32_t* d_env_mask, int flags,
                                          void* hip_stream) {
  if (!b || !d_draw) return fail("null argument");
  HIP_TRY(hipSetDevice(b->device));
  const HostModel& m = b->model->hm;
  const int nj = m.njnt;
  if (!b->d_rj_i) {
    std::vector<int> ji(3 * (size_t)std::max(1, nj));
    for (int j = 0; j < nj; j++) { ji[j] = m.jnt_type[j]; ji[nj + j] = m.jnt_qposadr[j]; ji[2*nj + j] = m.jnt_limited[j]; }
    HIP_TRY(hipMalloc((void**)&b->d_rj_i, ji.size() * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&b->d_rj_r, 2 * (size_t)std::max(1, nj) * sizeof(double)));
    HIP_TRY(hipMemcpy(b->d_rj_i, ji.data(), ji.size() * sizeof(int), hipMemcpyHostToDevice));
    if (nj) HIP_TRY(hipMemcpy(b->d_rj_r, m.jnt_range.data(), 2 * (size_t)nj * sizeof(double), hipMemcpyHostToDevice));
  }
  const dim3 grid((b->B + 255) / 256), block(256);
  void* q = find_field(b, "qpos")->dev;
  if (b->precision == 64)
    hipLaunchKernelGGL(randomize_joints_kernel<double>, grid, block, 0, (hipStream_t)hip_stream, (double*)q, b->B, nj, (const int*)b->d_rj_i,
                       (const double*)b->d_rj_r, (uint32_t)seed, (uint32_t)(seed >> 32), d_draw, d_env_mask, flags);
  else
    hipLaunchKernelGGL(randomize_joints_kernel<float>, grid, block, 0, (hipStream_t)hip_stream, (float*)q, b->B, nj, (const int*)b->d_rj_i,
                       (const double*)b->d_rj_r, (uint32_t)seed, (uint32_t)(seed >> 32), d_draw, d_env_mask, flags);
  HIP_TRY(hipGetLastError());
  // qpos was edited behind the stashes' back
  return dmc_batch_invalidate_async(b, hip_stream);
}

// ---- wave trace: when every wave of the LAST launch started / finished its item (100 MHz constant clock) -------
extern "C" int dmc_batch_wave_trace(dmc_batch* b, int enable, int32_t* dst, int* nitems) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->device));
  const int n = (b->B * b->geom.lpe + 63) / 64;
  if (nitems) *nitems = n;
  if (dst) {
    if (!b->d_trace) return fail("wave trace not enabled");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dst, b->d_trace, (size_t)64 * n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
  }
  HIP_TRY(hipDeviceSynchronize());
  if (b->d_trace) { (void)hipFree(b->d_trace); b->d_trace = nullptr; }
  b->trace_launch = 0;
  if (enable) {
    HIP_TRY(hipMalloc((void**)&b->d_trace, (size_t)64 * n * sizeof(int)));
    HIP_TRY(hipMemset(b->d_trace, 0, (size_t)64 * n * sizeof(int)));
  }
  return 0;
}

// ---- per-phase cycle profile (only meaningful in -DDMC_PROFILE builds) -----------
extern "C" int dmc_batch_prof_enable(dmc_batch* b, int enable) {
  if (!b) return fail("null batch");
#ifndef DMC_PROFILE
  if (enable) return fail("library built without DMC_PROFILE");
#endif
  HIP_TRY(hipSetDevice(b->device));
  if (b->d_prof) { (void)hipFree(b->d_prof); b->d_prof = nullptr; }
  if (!enable) return 0;
  HIP_TRY(hipMalloc((void**)&b->d_prof, (size_t)32 * b->B * sizeof(long long)));
  HIP_TRY(hipMemset(b->d_prof, 0, (size_t)32 * b->B * sizeof(long long)));
  return 0;
}
// dst: (PROF_N) mean cycles per env, accumulated since enable; returns PROF_N in *n
extern "C" int dmc_batch_prof_get(dmc_batch* b, double* dst, int* n) {
  if (!b || !dst || !n) return fail("null argument");
  if (!b->d_prof) return fail("profiling not enabled");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<long long> tmp((size_t)32 * b->B);
  HIP_TRY(hipMemcpy(tmp.data(), b->d_prof, tmp.size() * sizeof(long long), hipMemcpyDeviceToHost));
  for (int k = 0; k < 32; k++) { double s = 0; for (int e = 0; e < b->B; e++) s += (double)tmp[(size_t)k * b->B + e]; dst[k] = s / b->B; }
  *n = 32;
  return 0;
}
