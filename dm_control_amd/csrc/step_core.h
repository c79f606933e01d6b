// step_core.h -- the batched physics step, one environment per LPE-lane group.
//
// Replaces, for a whole batch at once, what the reference does one environment
// at a time through mujoco.mj_step / mj_step1 / mj_step2 / mj_forward
// (dm_control/mujoco/engine.py:147-176,335-343): kinematics + COM frame, CRB mass
// matrix + factorisation, collision, constraint assembly, velocity stage (RNE
// bias, passive), actuation, Newton constraint solve, semi-implicit Euler.
//
// Execution model (MI355X / gfx950): a group of LPE lanes of one 64-wide
// wavefront owns one environment.  Everything the group shares lives in LDS
// (`s`/`si` scratch, `mi`/`mr` model tables); lanes split loops as
// `for (i = lane; i < n; i += LPE)` and meet at DMC_WSYNC() (a wave-level
// fence: LDS ops of one wave retire in order, so no s_barrier is needed).
// Cross-lane arithmetic goes through group_sum/group_scan only, so the same
// source is valid for any LPE in {1,..,64}; tests/emu compiles it with LPE=1 on
// the host purely to unit-test the indexing logic without a GPU.
//
// Arithmetic order deliberately mirrors the fp64 oracle (oracle/mjstep_oracle.c)
// wherever a lane-parallel form with the identical operation order exists
// (tree accumulations in child-descending order, right-looking Cholesky,
// column-oriented substitution), so that the fp64 instantiation tracks the
// oracle to rounding noise; reductions over constraint rows are the exception.
#pragma once
#include <math.h>
#include <stdint.h>
#ifdef DMC_HOST_EMU
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#endif

#include "../../include/dmc_model_layout.h"
#include "step_layout.h"

#ifdef DMC_HOST_EMU
#define DMC_DEV inline
#define DMC_FN inline
#define DMC_LDS
#define DMC_GLB
// (host build: a counter -- how many wave-level fences one step executes is what a match on several waves would have to
// turn into workgroup barriers: scripts/multiwave_probe.py)
#define DMC_WSYNC() ((void)++dmc_emu_wsync_count)
static long long dmc_emu_wsync_count = 0;
#else
#define DMC_DEV __device__ __forceinline__
// out-of-line device functions (one copy of the code for all call sites) taking
// explicitly LDS-qualified pointers so that they still compile to ds_* ops
#define DMC_FN __device__ __attribute__((noinline, not_tail_called))
#define DMC_LDS __attribute__((address_space(3)))
#define DMC_GLB __attribute__((address_space(1)))   // the per-env global scratch: global_load / global_store, not flat
#define DMC_WSYNC()                                             \
  do {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      \
    __builtin_amdgcn_wave_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      \
  } while (0)
#endif

namespace dmc {
#ifdef DMC_HOST_EMU
inline int& emu_split_solves() { static int n = 0; return n; }
inline long long* emu_ls_counts() { static long long n[6] = {0, 0, 0, 0, 0, 0}; return n; }      // line searches, their cost evaluations, noslip passes, their sweeps, QCQP calls, their iterations      // solves whose H was taken as block diagonal over the trees (tests)
#endif

#if defined(DMC_PROFILE) && !defined(DMC_HOST_EMU)
// phase cycle counters in the env's LDS scratch (lane 0): the stages stay out of line, so the profiling build has the
// production kernel's structure (round 2's profiler kept its counters in the core object and had to inline the stages)
#define DMC_PROF(id) do { if (lane == 0) { const int t_ = (int)__builtin_readcyclecounter(); SI(prof)[id] += t_ - SI(prof)[33]; SI(prof)[33] = t_; } } while (0)
#else
#define DMC_PROF(id) ((void)0)
#endif
#ifdef DMC_TRACE_SUB
#define DMC_TSUB(region, cond, row) do { if (DMC_TRACE_SUB == (region) && (cond)) sub_stamp(row); } while (0)
#else
#define DMC_TSUB(region, cond, row) ((void)0)
#endif
enum { PROF_LOAD = 0, PROF_KIN, PROF_COM, PROF_CRB, PROF_COLL, PROF_CONSTR, PROF_COMVEL, PROF_RNE, PROF_SENS, PROF_ACT,
       PROF_ACC, PROF_SOL_INIT, PROF_SOL_GRAD, PROF_SOL_LS, PROF_SOL_UPD, PROF_EULER, PROF_TRAIL, PROF_STORE,
       PROF_NOSLIP, PROF_HESS, PROF_FACTOR, PROF_SOLVE, PROF_LS_SETUP, PROF_X1, PROF_X2, PROF_X3, PROF_X4, PROF_X5, PROF_X6,
       PROF_X7, PROF_X8, PROF_N };      // X1..X8: movable sub-markers of an investigation

// output selection bits (which derived arrays a launch writes back to HBM)
enum {
  OUT_SENSOR = 1 << 0, OUT_XPOS = 1 << 1, OUT_XQUAT = 1 << 2, OUT_XMAT = 1 << 3,
  OUT_XIPOS = 1 << 4, OUT_GEOM = 1 << 5, OUT_SITE = 1 << 6, OUT_SUBTREE_COM = 1 << 7,
  OUT_QACC = 1 << 8, OUT_ACTUATOR = 1 << 9, OUT_CONTACT = 1 << 10, OUT_QFRC = 1 << 11, OUT_CVEL = 1 << 12,
  OUT_CONTACT_IDS = 1 << 13,   // contact_geom1 / contact_geom2 only (contact scans of task layers)
  OUT_ALL = 0x7fffffff
};

// SoA device arrays: element (k, env) of a field lives at base[k * B + env]
template <typename T>
struct StepIO {
  int B;
  T *qpos, *qvel, *ctrl, *qacc_warmstart, *qfrc_applied;
  T *act;   // (na): activation states of actuators with dynamics
  double* time;   // always fp64: 1000 x 0.01 s must not drift in fp32 batches
  T *sensordata, *xpos, *xquat, *xmat, *xipos, *geom_xpos, *geom_xmat;
  T *site_xpos, *site_xmat, *subtree_com, *qacc, *actuator_force, *qfrc_actuator;
  T *qfrc_bias, *qfrc_constraint, *contact_dist, *contact_pos, *contact_frame;
  T *contact_force;   // (6 nconmax): mj_contactForce of the rows solved in this launch's last full pass
  T *cvel;            // (6 nbody): com-based body velocities, for mj_objectVelocity on the host
  int *ncon, *nefc, *solver_iter, *warning, *contact_geom1, *contact_geom2;
  // per-env override of a step launch: 0 = as launched, 1 = mj_forward with actuation disabled instead (an
  // environment that was just re-initialised: reset_context's after_reset), 2 = leave the environment untouched
  const int* env_mode;
  // Work queue of a launch whose grid is smaller than the batch (resident workgroups only): work[0] hands out the
  // next wave-sized item, work[1] counts finished waves (the last one re-arms both).  Null: one item per wave.
  int* work;
  // Longest-first scheduling of those launches: cost[item] = wave time (cycles / 64) the item took in the previous
  // launch, written by the step kernel; order[k] = the item handed out k-th, built from it before every launch
  // (order_kernel, dmc_api.hip).  Null: items in index order.
  int* cost; const int* order;
  // Sliced items (queued step launches of more than one physics step): the item's env-step is handed out as `slices`
  // pieces of consecutive physics steps, round by round -- piece s of every item before piece s + 1 of any -- so that
  // the launch ends within one PIECE of the last claim instead of within one whole env-step (a fallen 62-dof walker's
  // six substeps take 6 ms of an 8 ms launch, and what an item will cost is only half predictable from its last
  // launch).  Between pieces the state travels through the state arrays (nothing else is carried: every pass rebuilds
  // its derived arrays); prog[item] = pieces completed, zeroed by the launch's last wave.  slices <= 1 or prog == null:
  // whole items.  Between two pieces (qpos, qvel, qacc_warmstart, act, time) live in `hand`, an env-major record of
  // hand_n 8-byte words per env written with write-through (sc1) stores and read with loads that bypass the reading
  // CU's L1 (StepCore::store_handoff / load_handoff; MI355X_MICROARCH.md, inter-workgroup visibility: 8-byte agent-scope
  // atomics on both sides need no fence); the SoA state arrays are written by the last piece only.
  // nxcd: queues of a queued launch, one per XCD (step_kernel_body); 1 = a single queue.
  int* prog = nullptr; int slices = 0; int nxcd = 1;
  unsigned long long* hand = nullptr; int hand_n = 0;
  // Task epilogue (kernels built with -DDMC_TASK_HEADER: suite/fused_env.py): device copy of the generated task layer's
  // argument block; non-null: after an environment's step launch the wave that ran it evaluates that environment's
  // observation / reward / termination and, where the episode ended, writes the next start state -- the whole
  // control.Environment.step (rl/control.py:99-127) in one launch.  Ignored by kernels built without a task.
  const void* task_args = nullptr;
  // optional wave trace (dmc_batch_wave_trace): a ring of the last 8 launches, (8, 8, nitems) ints; per launch the rows
  // are the constant-rate clock (100 MHz) at which the item's wave entered the kernel (before the tables are staged),
  // started and finished the item, its workgroup index, and the clock after the opening position / velocity stage, the
  // (first) acceleration stage, the (first) integration and the trailing stage of a step launch; null: no trace
  int* trace; int trace_slot;
  // substep probe (dmc_batch_set_step_probe): the world position of ONE geom after every physics step of a legacy step
  // launch, (probe_cap, 3, B) -- what an after_substep hook that only looks at a position needs (the soccer goal /
  // out-of-court detectors, position_detector.py), so that a control step with such hooks is still one launch; null: off
  T* probe = nullptr; int probe_geom = 0, probe_cap = 0;
  // rollout mode: per-env-step inputs / outputs, (T, rows, B); any may be null
  const T* ctrl_seq; T *qpos_seq, *qvel_seq, *sensor_seq;
  // optional per-env stash of the position / velocity stage (what mjData keeps between the mj_step1 that ends one
  // legacy Physics.step() and the mj_step2 that begins the next): env-major, (B, n_keep) reals and (B, n_si + 4)
  // ints whose first word is the epoch the stash was written in (valid iff equal to *epoch).  The epoch lives in
  // device memory and is bumped by a kernel on the caller's stream (dmc_batch_invalidate_async), so an invalidation
  // recorded into a HIP graph is replayed with it -- a by-value epoch would be frozen at capture time.
  T* stash_r; int* stash_i; const int* epoch;
  // kinematic stash (on by default): what mj_kinematics / mj_comPos / mj_comVel derive from (qpos, qvel), kept per env
  // between legacy steps; (nq + nv + n_kin, ...) reals env-major and one epoch int per env (StepCore::load_kstash)
  T* kstash; int* kstash_i;
  long long* prof;   // optional (DMC_PROFILE builds): (PROF_N, B) cycle counters
  T* debug;      // optional: (n_sr, ndebug) dump of the env scratch after forward
  int* debug_i;  // optional: (n_si, ndebug)
  int ndebug;
};

#ifndef DMC_STATIC_FEATURES
#define DMC_STATIC_FEATURES 1
#endif
// ---------------------------------------------------------------------------
// scalar math (expression order mirrors the oracle)
// ---------------------------------------------------------------------------
template <typename T> DMC_DEV T t_sqrt(T x) { return (T)sqrt((double)x); }
template <> DMC_DEV float t_sqrt<float>(float x) { return sqrtf(x); }
// 1 / sqrt(x) for the Cholesky pivots.  fp32: the hardware reciprocal square root (v_rsq_f32, 1 ulp) instead of a
// correctly rounded sqrt followed by a correctly rounded division -- ~25 instructions less on the dependent chain of
// every column (a fifth of the 62 x 62 factorisation); fp64 keeps the exact sequence the oracle uses.
template <typename T> DMC_DEV T t_rsqrt(T x) { return 1 / t_sqrt(x); }
#if !defined(DMC_HOST_EMU) && !defined(DMC_EXACT_RSQ)
template <> DMC_DEV float t_rsqrt<float>(float x) { return __builtin_amdgcn_rsqf(x); }
#endif
template <typename T> DMC_DEV T t_sin(T x) { return (T)sin((double)x); }
template <> DMC_DEV float t_sin<float>(float x) { return sinf(x); }
template <typename T> DMC_DEV T t_cos(T x) { return (T)cos((double)x); }
template <> DMC_DEV float t_cos<float>(float x) { return cosf(x); }
template <typename T> DMC_DEV T t_pow(T x, T y) { return (T)pow((double)x, (double)y); }
template <> DMC_DEV float t_pow<float>(float x, float y) { return powf(x, y); }
template <typename T> DMC_DEV T t_exp(T x) { return (T)exp((double)x); }
template <typename T> DMC_DEV T t_atan2(T y, T x) { return (T)atan2((double)y, (double)x); }
template <> DMC_DEV float t_atan2<float>(float y, float x) { return atan2f(y, x); }
template <typename T> DMC_DEV T t_fmod(T x, T y) { return (T)fmod((double)x, (double)y); }
template <> DMC_DEV float t_fmod<float>(float x, float y) { return fmodf(x, y); }
template <> DMC_DEV float t_exp<float>(float x) { return expf(x); }
// Correctly rounded fp32 division / square root whatever the build's fp32 division mode (step_kernels_f32 is compiled
// with the 2.5-ulp hardware forms): for the few places whose branch decisions sit on an absolute 1e-10 (the PGS block
// updates) and are not on the hot path of any BASELINE configuration.
template <typename T> DMC_DEV T t_div_exact(T a, T b) { return a / b; }
template <> DMC_DEV float t_div_exact<float>(float a, float b) { return (float)((double)a / (double)b); }
template <typename T> DMC_DEV T t_sqrt_exact(T x) { return (T)sqrt((double)x); }
template <typename T> DMC_DEV T t_abs(T x) { return x < 0 ? -x : x; }
template <typename T> DMC_DEV T t_max(T a, T b) { return a > b ? a : b; }
template <typename T> DMC_DEV T t_min(T a, T b) { return a < b ? a : b; }
template <typename T> DMC_DEV bool t_bad(T x) { return !(x == x) || x > (T)DMC_MAXVAL || x < -(T)DMC_MAXVAL; }

template <typename T> DMC_DEV T dot3(const T* a, const T* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
template <typename T> DMC_DEV void cross3(T* r, const T* a, const T* b) {
  T t0 = a[1]*b[2] - a[2]*b[1], t1 = a[2]*b[0] - a[0]*b[2], t2 = a[0]*b[1] - a[1]*b[0];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
template <typename T> DMC_DEV T normalize3(T* v) {
  T n = t_sqrt(dot3(v, v));
  if (n < (T)DMC_MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; }
  else { T s = 1 / n; v[0] *= s; v[1] *= s; v[2] *= s; }
  return n;
}
template <typename T> DMC_DEV void normalize4(T* q) {
  T n = t_sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < (T)DMC_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else if (t_abs(n - 1) > (T)DMC_MINVAL) { T s = 1 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
template <typename T> DMC_DEV void mul_quat(T* r, const T* a, const T* b) {
  T t0 = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  T t1 = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  T t2 = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  T t3 = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
template <typename T> DMC_DEV void quat2mat(T* m, const T* q) {
  T q00 = q[0]*q[0], q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
  T q11 = q[1]*q[1], q12 = q[1]*q[2], q13 = q[1]*q[3];
  T q22 = q[2]*q[2], q23 = q[2]*q[3], q33 = q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2*(q12 - q03); m[2] = 2*(q13 + q02);
  m[3] = 2*(q12 + q03); m[5] = 2*(q23 - q01);
  m[6] = 2*(q13 - q02); m[7] = 2*(q23 + q01);
}
template <typename T> DMC_DEV void mul_mat_vec3(T* r, const T* m, const T* v) {
  T t0 = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  T t1 = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  T t2 = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
template <typename T> DMC_DEV void mul_matT_vec3(T* r, const T* m, const T* v) {
  T t0 = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  T t1 = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  T t2 = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
template <typename T> DMC_DEV void rot_vec_quat(T* r, const T* v, const T* q) {
  T m[9]; quat2mat(m, q); mul_mat_vec3(r, m, v);
}
template <typename T> DMC_DEV void axisangle2quat(T* q, const T* axis, T angle) {
  // Straight-line on purpose: angle == 0 gives s = 0, c = 1, i.e. the identity MuJoCo
  // returns early with; an early-out branch (or sincos()'s pointer outputs) makes the
  // compiler route q through scratch memory.
  const T s = t_sin(angle * (T)0.5), c = t_cos(angle * (T)0.5);
  q[0] = c; q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
template <typename T> DMC_DEV void quat_integrate(T* quat, const T* vel, T scale) {
  T tmp[3] = {vel[0], vel[1], vel[2]}, qrot[4];
  T angle = scale * normalize3(tmp);
  axisangle2quat(qrot, tmp, angle);
  normalize4(quat);
  mul_quat(quat, quat, qrot);
}
template <typename T> DMC_DEV void inert_com(T* res, const T* inert, const T* mat, const T* dif, T mass) {
  T tmp[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) tmp[3*r + c] = mat[3*r + c] * inert[c];
  res[0] = tmp[0]*mat[0] + tmp[1]*mat[1] + tmp[2]*mat[2];
  res[1] = tmp[3]*mat[3] + tmp[4]*mat[4] + tmp[5]*mat[5];
  res[2] = tmp[6]*mat[6] + tmp[7]*mat[7] + tmp[8]*mat[8];
  res[3] = tmp[0]*mat[3] + tmp[1]*mat[4] + tmp[2]*mat[5];
  res[4] = tmp[0]*mat[6] + tmp[1]*mat[7] + tmp[2]*mat[8];
  res[5] = tmp[3]*mat[6] + tmp[4]*mat[7] + tmp[5]*mat[8];
  res[0] += mass * (dif[1]*dif[1] + dif[2]*dif[2]);
  res[1] += mass * (dif[0]*dif[0] + dif[2]*dif[2]);
  res[2] += mass * (dif[0]*dif[0] + dif[1]*dif[1]);
  res[3] -= mass * dif[0]*dif[1];
  res[4] -= mass * dif[0]*dif[2];
  res[5] -= mass * dif[1]*dif[2];
  res[6] = mass*dif[0]; res[7] = mass*dif[1]; res[8] = mass*dif[2];
  res[9] = mass;
}
template <typename T> DMC_DEV void mul_inert_vec(T* res, const T* i, const T* v) {
  res[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  res[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  res[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  res[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  res[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  res[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
template <typename T> DMC_DEV void cross_motion(T* res, const T* vel, const T* v) {
  res[0] = -vel[2]*v[1] + vel[1]*v[2];
  res[1] =  vel[2]*v[0] - vel[0]*v[2];
  res[2] = -vel[1]*v[0] + vel[0]*v[1];
  res[3] = -vel[2]*v[4] + vel[1]*v[5];
  res[4] =  vel[2]*v[3] - vel[0]*v[5];
  res[5] = -vel[1]*v[3] + vel[0]*v[4];
  res[3] += -vel[5]*v[1] + vel[4]*v[2];
  res[4] +=  vel[5]*v[0] - vel[3]*v[2];
  res[5] += -vel[4]*v[0] + vel[3]*v[1];
}
template <typename T> DMC_DEV void cross_force(T* res, const T* vel, const T* f) {
  res[0] = -vel[2]*f[1] + vel[1]*f[2];
  res[1] =  vel[2]*f[0] - vel[0]*f[2];
  res[2] = -vel[1]*f[0] + vel[0]*f[1];
  res[3] = -vel[2]*f[4] + vel[1]*f[5];
  res[4] =  vel[2]*f[3] - vel[0]*f[5];
  res[5] = -vel[1]*f[3] + vel[0]*f[4];
  res[0] += -vel[5]*f[4] + vel[4]*f[5];
  res[1] +=  vel[5]*f[3] - vel[3]*f[5];
  res[2] += -vel[4]*f[3] + vel[3]*f[4];
}
template <typename T> DMC_DEV T dot_n(const T* a, const T* b, int n) {
  T s = 0; for (int i = 0; i < n; i++) s += a[i]*b[i]; return s;
}

// ---------------------------------------------------------------------------
// group primitives (LPE lanes of one wave)
// ---------------------------------------------------------------------------
#ifndef DMC_HOST_EMU
// DPP cross-lane moves inside a 16-lane row (no LDS traffic, ~VALU latency)
template <int CTRL> DMC_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> DMC_DEV float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL> DMC_DEV double dpp_f(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = dpp_i<CTRL>((int)(b & 0xffffffffll)), hi = dpp_i<CTRL>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#endif
// Exchanges across the 16-lane rows / the 32-lane halves of a wave with the gfx950 row / half swaps
// (v_permlane16_swap / v_permlane32_swap: VALU moves) instead of a trip through the LDS crossbar (ds_bpermute, what
// __shfl_xor compiles to): swapping a value with itself leaves {even row's copy, odd row's copy} of each row pair in
// the two results, whose sum / max is the same in both rows.  Reductions sit on the critical path of every solver
// iteration (their results feed the next branch), ~25 per Newton iteration.
#ifndef DMC_HOST_EMU
struct Pair32 { unsigned a, b; };
DMC_DEV Pair32 swap16(unsigned x) { const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); Pair32 p = {r[0], r[1]}; return p; }
DMC_DEV Pair32 swap32(unsigned x) { const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); Pair32 p = {r[0], r[1]}; return p; }
template <int W> DMC_DEV Pair32 swapW(unsigned x) { return W == 16 ? swap16(x) : swap32(x); }
template <int W> DMC_DEV float xsum(float v) { const Pair32 p = swapW<W>(__float_as_uint(v)); return __uint_as_float(p.a) + __uint_as_float(p.b); }
template <int W> DMC_DEV int xsum(int v) { const Pair32 p = swapW<W>((unsigned)v); return (int)p.a + (int)p.b; }
template <int W> DMC_DEV double xsum(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const Pair32 lo = swapW<W>((unsigned)u), hi = swapW<W>((unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi.a << 32) | lo.a) + __builtin_bit_cast(double, ((unsigned long long)hi.b << 32) | lo.b);
}
template <int W> DMC_DEV int xmax(int v) { const Pair32 p = swapW<W>((unsigned)v); return (int)p.a > (int)p.b ? (int)p.a : (int)p.b; }
#endif
// Sum over the LPE lanes of a group; every lane receives the total.  Same
// pairing tree as an xor butterfly (1, 2, 4, 8 inside a row via DPP quad_perm /
// row_half_mirror / row_mirror, then 16 and 32 via the row / half swaps).
template <int LPE, typename V> DMC_DEV V group_sum(V v) {
#ifndef DMC_HOST_EMU
  if (LPE >= 2) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  if (LPE >= 4) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  if (LPE >= 8) v += dpp_f<0x141>(v);   // row_half_mirror
  if (LPE >= 16) v += dpp_f<0x140>(v);  // row_mirror
  if (LPE >= 32) v = xsum<16>(v);
  if (LPE >= 64) v = xsum<32>(v);
#endif
  return v;
}
// value held by lane `k` of each group when k is WAVE-uniform (a loop counter): v_readlane
// into an SGPR (one per group of the wave) instead of a trip through the LDS crossbar
#ifndef DMC_HOST_EMU
DMC_DEV float readlane_t(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
DMC_DEV int readlane_t(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
DMC_DEV double readlane_t(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#endif
template <int LPE, typename V> DMC_DEV V wave_bcast(V v, int k) {
#ifndef DMC_HOST_EMU
  if (LPE == 64) return readlane_t(v, k);
  const int g = (int)(__lane_id()) / LPE;   // which group of the wave this lane belongs to
  V r = readlane_t(v, k);
#pragma unroll
  for (int q = 1; q < 64 / LPE; q++) { const V w = readlane_t(v, q*LPE + k); r = g == q ? w : r; }
  return r;
#else
  (void)k; return v;
#endif
}
// The same when every lane that holds something sits in the FIRST 16-lane row of its group (one lane per matrix row of a
// model with nv <= 16): the gfx90a+ DPP control row_newbcast:k hands lane k of each 16-lane row to all lanes of that row
// in ONE VALU move (folded into the consuming multiply where the encoding allows) -- against two v_readlane, a trip
// through two SGPRs and a v_cndmask per value for two environments per wave.  The 9 x 9 factorisations of the cheetah
// were ~200 instructions of which 135 were these broadcasts; the lanes of the group's other rows receive the value of
// THEIR row's lane k, which nothing reads (they own no matrix row).  k must be a constant after unrolling.
#ifndef DMC_HOST_EMU
template <int CTRL> DMC_DEV float dpp_all(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
template <int CTRL> DMC_DEV double dpp_all(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (every lane of a row_newbcast has a source lane: bound_ctrl spares the move that would initialise the "old" value)
template <typename V> DMC_DEV V row_bcast16(V v, int k) {
  switch (k & 15) {
    case 0: return dpp_all<0x150>(v); case 1: return dpp_all<0x151>(v); case 2: return dpp_all<0x152>(v); case 3: return dpp_all<0x153>(v);
    case 4: return dpp_all<0x154>(v); case 5: return dpp_all<0x155>(v); case 6: return dpp_all<0x156>(v); case 7: return dpp_all<0x157>(v);
    case 8: return dpp_all<0x158>(v); case 9: return dpp_all<0x159>(v); case 10: return dpp_all<0x15A>(v); case 11: return dpp_all<0x15B>(v);
    case 12: return dpp_all<0x15C>(v); case 13: return dpp_all<0x15D>(v); case 14: return dpp_all<0x15E>(v); default: return dpp_all<0x15F>(v);
  }
}
#endif
// broadcast of matrix row k's value among the N <= LPE row-holding lanes of a group
template <int LPE, int N, typename V> DMC_DEV V bcast_rows(V v, int k) {
#if !defined(DMC_HOST_EMU) && !defined(DMC_NO_ROW_NEWBCAST)
  if constexpr (N <= 16 && LPE >= 16) return row_bcast16(v, k);
#endif
  return wave_bcast<LPE>(v, k);
}
template <int LPE> DMC_DEV int group_max(int v) {
#ifndef DMC_HOST_EMU
  int w;
  if (LPE >= 2) { w = dpp_i<0xB1>(v); v = w > v ? w : v; }
  if (LPE >= 4) { w = dpp_i<0x4E>(v); v = w > v ? w : v; }
  if (LPE >= 8) { w = dpp_i<0x141>(v); v = w > v ? w : v; }
  if (LPE >= 16) { w = dpp_i<0x140>(v); v = w > v ? w : v; }
  if (LPE >= 32) v = xmax<16>(v);
  if (LPE >= 64) v = xmax<32>(v);
#endif
  return v;
}
// exclusive prefix sum over the group; *total receives the group sum.  Hillis-Steele inside a 16-lane row with DPP
// row shifts (zeros shifted in), then the row totals travel with row_bcast:15 / row_bcast:31 -- no LDS crossbar trips
// (__shfl_up is a ds_bpermute: six dependent ones per scan).
template <int LPE> DMC_DEV int group_scan(int v, int lane, int* total) {
#ifndef DMC_HOST_EMU
  int inc = v;
  (void)lane;
  if (LPE >= 2) inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);    // row_shr:1
  if (LPE >= 4) inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);    // row_shr:2
  if (LPE >= 8) inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);    // row_shr:4
  if (LPE >= 16) inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);   // row_shr:8
  if (LPE >= 32) inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1, 3
  if (LPE >= 64) inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2, 3
  *total = wave_bcast<LPE>(inc, LPE - 1);
  return inc - v;
#else
  (void)lane; *total = v; return 0;
#endif
}

// ---------------------------------------------------------------------------
// out-of-line LDS routines shared by several call sites
// ---------------------------------------------------------------------------
// Symmetric n x n matrices (the factor of M, H = M + J'DJ and its factor) are stored as their lower
// triangle packed BY COLUMNS: entry (i, j), i >= j, lives at tri_c0(j, n) + i - j.  Column k is
// contiguous, and the entries a right-looking Cholesky still has to touch at step k are a suffix.
DMC_DEV int tri_c0(int j, int n) { return j*n - ((j*(j - 1)) >> 1); }
DMC_DEV int tri_at(int i, int j, int n) { return tri_c0(j, n) + i - j; }
// inverse of tri_at for a packed index t of an m x m triangle: column j and row i
DMC_DEV void tri_unrank(int t, int m, int* i, int* j) {
  const float b = (float)(2*m + 1);
  int c = (int)((b - sqrtf(b*b - 8.0f*(float)t)) * 0.5f);
  c = c < 0 ? 0 : (c > m - 1 ? m - 1 : c);
  if (tri_c0(c, m) > t) c--;
  else if (c + 1 < m && tri_c0(c + 1, m) <= t) c++;
  *j = c; *i = c + (t - tri_c0(c, m));
}
// In-place Cholesky of a packed lower triangle, same operation order as the oracle.  On exit: strict
// lower part = L, diagonal = 1/L[k][k].  Two wave fences per column: scale column k, then every
// remaining entry (i, j), j > k, is updated by one lane with  A[i][j] -= L[i][k] L[j][k].
// a - b c: in the fp32 kernels ONE fused operation, said explicitly -- left to the contraction pass, the SLP vectoriser first
// pairs the products of neighbouring columns into v_pk_mul_f32 and the fusion is lost (two moves, a packed product and two
// subtractions where two FMAs do); in fp64 the two roundings of the oracle.
// (FUSE: the small row routines, N <= 16; the larger ones keep the expression the compiler has always seen -- on the 27-dof
// model the explicit form bought nothing and moved the mean iteration count, profiles/r05_s7_ab_large_models.log)
template <bool FUSE, typename T> DMC_DEV T nmsub(T a, T b, T c) {
#ifndef DMC_HOST_EMU
  if constexpr (FUSE && sizeof(T) == 4) return __builtin_fmaf(-b, c, a);
#endif
  return a - b * c;
}
#ifndef DMC_HOST_EMU
// Row-per-lane factor -> the packed triangle (column j at tri_c0(j, N), rows j .. N-1): lane i holds (i, j) for j <= i.
// Stored WITHOUT a predicate per column: the columns go out last to first, and a lane above the diagonal (i < j) aims
// its don't-care value at tri_c0(j, N) + i - j -- a slot of an EARLIER column (>= 0 because tri_c0(j, N) >= j), which that
// column's own store, issued later by the same wave, overwrites with the entry that belongs there.  One exec mask for
// the N row-holding lanes instead of a compare / mask / branch / restore sequence per column (9 x 8 instructions on the
// 9-dof model, a quarter of the factorisation).
template <typename T, int N> DMC_DEV void store_factor_rows(DMC_LDS T* A, const T* a, int lane) {
  if (lane < N) {
#pragma unroll
    for (int j = N - 1; j >= 0; j--) { A[tri_c0(j, N) + lane - j] = a[j]; asm volatile("" ::: "memory"); }      // (in THIS order: to one lane the nine addresses are unrelated)
  }
}
#endif
template <typename T, int LPE>
DMC_FN void chol_factor_lds(DMC_LDS T* A, int n, int lane) {
  const int ntri = (n*(n + 1)) >> 1;
  for (int k = 0; k < n; k++) {
    DMC_WSYNC();
    const int ck = tri_c0(k, n);
    T akk = A[ck];
    if (akk < (T)DMC_MINVAL) akk = (T)DMC_MINVAL;
    const T inv = t_rsqrt(akk);
    for (int i = k + 1 + lane; i < n; i += LPE) A[ck + i - k] *= inv;
    DMC_WSYNC();
    if (lane == 0) A[ck] = inv;
    const int c1 = ck + n - k, m = n - k - 1;   // trailing (n-k-1) x (n-k-1) triangle starts at c1
    for (int t = lane; t < ntri - c1; t += LPE) {
      int ii, jj;
      tri_unrank(t, m, &ii, &jj);
      A[c1 + t] -= A[ck + 1 + ii] * A[ck + 1 + jj];
    }
  }
  DMC_WSYNC();
}
// Model-specialised kernels know nv at compile time: lane i of the group keeps row i of the
// matrix in N registers, pivots and scaled columns travel by v_readlane -- no LDS round trip
// and no fence per column (N = 27: ~1.1 k instructions instead of 27 fenced LDS sweeps).
// Same arithmetic per entry, in the same order, as chol_factor_lds: identical results.
#ifndef DMC_HOST_EMU
template <typename T, int LPE, int N>
DMC_FN void chol_factor_rows(DMC_LDS T* A, int lane) {
  static_assert(N >= 1 && N <= LPE, "one lane per matrix row");
  DMC_WSYNC();
  T a[N];
  const bool own = lane < N;
#pragma unroll
  for (int j = 0; j < N; j++) {      // (N <= 16: unpredicated loads, the value selected afterwards -- a predicated load costs an exec-mask round trip each)
    if constexpr (N > 16) a[j] = (own && j <= lane) ? A[tri_c0(j, N) + lane - j] : (T)0;      // (27 dofs: 3 % faster predicated -- half the reads)
    else {
      const int i_ = own && j <= lane ? lane : j;
      const T v = A[tri_c0(j, N) + i_ - j];
      a[j] = (own && j <= lane) ? v : (T)0;
    }
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    T akk = bcast_rows<LPE, N>(a[k], k);
    if (akk < (T)DMC_MINVAL) akk = (T)DMC_MINVAL;
    const T inv = t_rsqrt(akk);
    const T lik = a[k] * inv;
#pragma unroll
    for (int j = k + 1; j < N; j++) { const T ljk = bcast_rows<LPE, N>(lik, j); a[j] = nmsub<(N <= 16)>(a[j], lik, ljk); }
    a[k] = lane == k ? inv : lik;
  }
  store_factor_rows<T, N>(A, a, lane);
  DMC_WSYNC();
}
#endif
// The same factorisation on the MATRIX CORES for the large fp32 models (32 < N <= 64, one environment per wave): blocked
// right-looking U'U on 16 x 16 tiles held in the accumulator layout of v_mfma_f32_16x16x4_f32 -- lane 16 g + c, register r
// of a tile = its element (4 g + r, c).  Fed as BOTH operands, register by register, two tiles X, Y in that layout give
// X'Y (operand A reads lane l as A[l & 15][l >> 4], operand B as B[l >> 4][l & 15]: register r of X is X'[i][4 k + r],
// register r of Y is Y[4 k + r][j], and the four instructions r = 0 .. 3 cover the sixteen k) -- which is the trailing update
// A_ij -= U_ki' U_kj of the upper-triangular form, in place, with no layout conversion: 40 matrix instructions do what
// 1 891 v_readlane + v_fma pairs do in chol_factor_rows<62>.  The sixteen columns of a diagonal tile are eliminated on
// the vector ALU, together with the rest of their block row (which is the panel solve): the pivot comes by v_readlane,
// the column below it by DPP row_newbcast (the diagonal tile is kept whole and symmetric, so column C0 of a row group is
// lane C0 of that row group), the scaled pivot row reaches the other row groups through one ds_bpermute per tile; rows at
// or above the pivot get a zero multiplier instead of a predicate.  Rows / columns N .. 63 enter as the identity.  The
// packed triangle leaves as chol_factor_rows leaves it (scaled columns, 1 / L_kk on the diagonal); the sums run in another
// order (the products of a tile update are added k-slot by k-slot), so the factor differs from chol_factor_rows' by
// rounding.  Measured (scripts/chol_mfma_probe.hip, profiles/r06_chol_mfma_probe.log): 2 191 instructions against 5 118,
// 17.0 k cycles per factorisation against 38.9 k with five waves per CU.
#if !defined(DMC_HOST_EMU)
typedef float dmc_f4 __attribute__((ext_vector_type(4)));
template <int N> struct CholTiles {
  static constexpr int NB = (N + 15) / 16;
  struct LaneInfo { int g, col4, lane; float fgt[3]; };      // fgt[q] = 1 where the lane's row group g > q, else 0
  // the value of the lane of the same column in row group GC, for every row group: on the diagonal tile (the critical
  // chain) by two VALU swaps, on the rest of the block row through the LDS crossbar (off the chain; 18.5 k -> 15.3 k cycles)
  template <int GC, bool DIAG> static DMC_DEV float bcast_rowgroup(float x, int col4) {
    if constexpr (!DIAG) return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(col4 + 64 * GC, __builtin_bit_cast(int, x)));
    else {
      const unsigned u = __builtin_bit_cast(unsigned, x);
      const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // h[0]: row groups (0 1 0 1), h[1]: (2 3 2 3)
      const unsigned y = GC < 2 ? h[0] : h[1];
      const auto q = __builtin_amdgcn_permlane16_swap(y, y, false, false);      // q[0]: the even group everywhere, q[1]: the odd
      return __builtin_bit_cast(float, (GC & 1) ? q[1] : q[0]);
    }
  }
  template <int K, int C0> static DMC_DEV void eliminate_column(dmc_f4 (&t)[NB][NB], const LaneInfo& tl) {
    constexpr int GC = C0 >> 2, RC = C0 & 3;
    dmc_f4& D = t[K][K];
    const float inv = __builtin_amdgcn_rsqf(__builtin_amdgcn_fmed3f(readlane_t(D[RC], 16 * GC + C0), (float)DMC_MINVAL, __builtin_inff()));
    // the column below the pivot, scaled; rows at or above the pivot row get a zero multiplier, folded into the scale
    // (4 g + r > C0  <=>  r > RC ? g >= GC : g > GC)
    const float inv_ge = GC == 0 ? inv : inv * tl.fgt[GC > 0 ? GC - 1 : 0];
    const float inv_gt = GC == 3 ? 0.f : inv * tl.fgt[GC < 3 ? GC : 0];
    float ui[4];
#pragma unroll
    for (int r = 0; r < 4; r++) ui[r] = (GC == 3 && r <= RC) ? 0.f : dpp_all<0x150 + C0>(D[r]) * (r > RC ? inv_ge : inv_gt);
    const float scale = tl.g == GC ? inv : 1.f;      // the pivot row itself is scaled in place
#pragma unroll
    for (int j = K; j < NB; j++) {
      dmc_f4& P = t[K][j];
      P[RC] = P[RC] * scale;
      const float X = j == K ? bcast_rowgroup<GC, true>(P[RC], tl.col4) : bcast_rowgroup<GC, false>(P[RC], tl.col4);
#pragma unroll
      for (int r = 0; r < 4; r++) if (!(GC == 3 && r <= RC)) P[r] = P[r] - ui[r] * X;
    }
    D[RC] = (tl.lane == 16 * GC + C0) ? inv : D[RC];      // the packed form keeps 1 / L_kk on the diagonal
  }
  template <int K, int C0> struct Columns {
    static DMC_DEV void run(dmc_f4 (&t)[NB][NB], const LaneInfo& tl) {
      if constexpr (16 * K + C0 < N) eliminate_column<K, C0>(t, tl);      // (the columns past N are the identity's)
      if constexpr (C0 + 1 < 16) Columns<K, C0 + 1>::run(t, tl);
    }
  };
  template <int K> static DMC_DEV void block_column(dmc_f4 (&t)[NB][NB], const LaneInfo& tl) {
    Columns<K, 0>::run(t, tl);
#pragma unroll
    for (int i = K + 1; i < NB; i++) {
      const dmc_f4 nx = -t[K][i];
#pragma unroll
      for (int j = i; j < NB; j++) {
#pragma unroll
        for (int r = 0; r < 4; r++) t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(nx[r], t[K][j][r], t[i][j], 0, 0, 0);
      }
    }
    if constexpr (K + 1 < NB) block_column<K + 1>(t, tl);
  }
  static DMC_DEV constexpr int c0(int j) { return j * N - ((j * (j - 1)) >> 1); }      // tri_c0(j, N)
  // Packed index of element (R, C), R <= C, of tile (bi, bj), register r, for the lane (g, c): tri_c0(R) + C - R with
  // R = R0 + G (R0 = 16 bi + r, G = 4 g)  =  [tri_c0(R0) + 16 bj - R0] + [tri_c0(G) - G + c] - R0 G
  static DMC_DEV void factor(DMC_LDS float* A, int lane) {
    static_assert(N > 32 && N <= 64, "three or four tiles a side (store()'s spare slot is an entry of column 16 + c: N >= 32; below 33 dofs the row form is as fast)");
    const int g = lane >> 4, c = lane & 15, G = 4 * g;
    LaneInfo tl; tl.g = g; tl.col4 = 4 * c; tl.lane = lane;
#pragma unroll
    for (int q = 0; q < 3; q++) tl.fgt[q] = g > q ? 1.f : 0.f;
    const int up = c0(G) - G + c;                      // lane part of the upper-triangle index
    const int tc = ((c * (2 * N + 1 - c)) >> 1) - c;   // tri_c0(c) - c: lane part of the mirrored (lower-triangle) index
    dmc_f4 t[NB][NB];
#pragma unroll
    for (int bi = 0; bi < NB; bi++)
#pragma unroll
      for (int bj = bi; bj < NB; bj++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int R0 = 16 * bi + r;
          int at = c0(R0) + 16 * bj - R0 + up - R0 * G;
          if (bi == bj) {      // the diagonal tiles enter whole (symmetric): below the diagonal the mirrored entry
            const int low = c0(16 * bi) + tc - 16 * bi * c + R0 - 16 * bi + G;
            at = (G + r <= c) ? at : low;
          }
          float v = A[at];      // (unpredicated: an index past the triangle reads a neighbouring array, the value is dropped)
          if (16 * bj + 15 >= N) { const bool in = (16 * bj + c < N) && (bi < bj || 16 * bi + G + r < N); v = in ? v : ((bi == bj && G + r == c) ? 1.f : 0.f); }
          t[bi][bj][r] = v;
        }
    block_column<0>(t, tl);
    // stores: an entry that does not exist (below the diagonal of a diagonal tile, past column N) aims at the slot of the
    // lane's entry (row 4 g, column 16 + c) -- which exists for every lane and is stored LAST, over whatever landed there
    const int safe = 16 + up;
#pragma unroll
    for (int bi = NB - 1; bi >= 0; bi--)
#pragma unroll
      for (int bj = NB - 1; bj >= bi; bj--)
#pragma unroll
        for (int r = 3; r >= 0; r--) {
          const int R0 = 16 * bi + r;
          int at = c0(R0) + 16 * bj - R0 + up - R0 * G;
          bool ok = true;
          if (bi == bj) ok = G + r <= c;
          if (16 * bj + 15 >= N) ok = ok && (16 * bj + c < N);
          if (bi == bj || 16 * bj + 15 >= N) at = ok ? at : safe;
          if (!(bi == 0 && bj == 1 && r == 0)) A[at] = t[bi][bj][r];
        }
    asm volatile("" ::: "memory");
    A[safe] = t[0][1][0];
  }
};
template <int LPE, int N>
DMC_FN void chol_factor_tiles(DMC_LDS float* A, int lane) {
  static_assert(LPE == 64, "one environment per wave");
  DMC_WSYNC();
  CholTiles<N>::factor(A, lane);
  DMC_WSYNC();
}
#endif
// Substitution for model-specialised kernels: lane i loads its row and its column of L up
// front (all loads in flight together), then both sweeps run on registers and v_readlane --
// no LDS access inside the 2 N dependent steps.  Same operations as chol_solve_lds.
#ifndef DMC_HOST_EMU
template <typename T, int LPE, int N>
DMC_FN void chol_solve_rows(DMC_LDS T* x, const DMC_LDS T* Lm, const DMC_LDS T* b, int lane) {
  static_assert(N >= 1 && N <= LPE, "one lane per unknown");
  const int i = lane;
  const bool own = i < N;
  const int ci = tri_c0(own ? i : 0, N);
  T row[N], col[N];
#pragma unroll
  for (int k = 0; k < N; k++) {      // (N <= 16: unpredicated loads from in-range addresses, the values selected afterwards)
    if constexpr (N > 16) { row[k] = (own && k < i) ? Lm[tri_c0(k, N) + i - k] : (T)0; col[k] = (own && k > i) ? Lm[ci + k - i] : (T)0; }
    else {
      const T r_ = Lm[tri_c0(k, N) + ((own && k < i) ? i - k : 0)], c_ = Lm[ci + ((own && k > i) ? k - i : 0)];
      row[k] = (own && k < i) ? r_ : (T)0; col[k] = (own && k > i) ? c_ : (T)0;
    }
  }
  const T dinv_ = Lm[ci], b_ = b[own ? i : 0];
  const T dinv = own ? dinv_ : (T)0;      // 1 / L[i][i]
  T sreg = own ? b_ : (T)0;
  // Step k needs x_k = s_k / L[k][k]: every lane forms its own s_i * dinv_i (one VALU op, and lane k's is the value),
  // ONE cross-lane read fetches it, and the update is an unconditional FMA -- row[k] / col[k] are zeros where the
  // step does not reach, so lane k keeps its finished s_k and its x_k is s_k * dinv_k again after the loop.  Three
  // instructions per step instead of ~20 (two broadcasts, a scalar product moved back to a VGPR, a write-lane and two
  // predicated updates: 2 603 instructions for N = 62, 8 % of the 62-dof step).  The same products and differences as
  // before, bit for bit.
#pragma unroll
  for (int k = 0; k < N; k++) { const T xk = bcast_rows<LPE, N>(sreg*dinv, k); sreg = nmsub<(N <= 16)>(sreg, row[k], xk); }
  sreg = sreg*dinv;
#pragma unroll
  for (int k = N - 1; k >= 0; k--) { const T xk = bcast_rows<LPE, N>(sreg*dinv, k); sreg = nmsub<(N <= 16)>(sreg, col[k], xk); }
  if (own) x[i] = sreg*dinv;
  DMC_WSYNC();
}
#endif
// ---- block diagonal over the kinematic trees (StepDims::treemax) ------------------------------------------------
// M -- and H = M + J'DJ as long as no constraint row moves two trees -- is block diagonal over the kinematic trees of a
// multi-body scene (soccer 2v2: five trees of six dofs).  The row-per-lane routines above run the N columns one after
// the other, N (N + 1) / 2 cross-lane broadcasts + FMAs, of which all but the in-tree ones multiply exact zeros; a lone
// wave per SIMD pays ~9 cycles per instruction, so the 30 x 30 factorisation was 16 k cycles, 8 % of the soccer step, and
// each substitution 11 k.  Here every tree eliminates ITS column kk = 0 .. TM-1 at the same time: the pivot lane differs
// per tree, so the broadcasts are per-lane-addressed (ds_bpermute) instead of v_readlane: TM (TM + 1) / 2 of them for
// the whole matrix.  Entry for entry the same operations in the same order as chol_factor_rows / chol_solve_rows --
// what is skipped is  a - 0 * x -- so the results are bit-identical.  t0 / t1: first dof / 1 + last dof of the lane's
// tree.  One environment per wave (LPE = 64).
#ifndef DMC_HOST_EMU
DMC_DEV float lane_read(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v))); }
DMC_DEV double lane_read(double v, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename T, int LPE, int N, int TM>
DMC_FN void chol_factor_trees(DMC_LDS T* A, int lane, int t0, int t1) {
  static_assert(LPE == 64 && N <= LPE, "one lane per matrix row, one environment per wave");
  DMC_WSYNC();
  T a[TM];
  const bool own = lane < N;
#pragma unroll
  for (int kk = 0; kk < TM; kk++) { const int j = t0 + kk; a[kk] = (own && j <= lane) ? A[tri_c0(j, N) + lane - j] : (T)0; }
#pragma unroll
  for (int kk = 0; kk < TM; kk++) {
    const int src = t0 + kk;
    T akk = lane_read(a[kk], src);
    if (akk < (T)DMC_MINVAL) akk = (T)DMC_MINVAL;
    const T inv = t_rsqrt(akk);
    const T lik = (own && src < t1) ? a[kk] * inv : (T)0;      // (a tree with fewer than TM dofs sits these columns out)
#pragma unroll
    for (int jj = kk + 1; jj < TM; jj++) { const T ljk = lane_read(lik, t0 + jj); a[jj] = a[jj] - lik * ljk; }
    a[kk] = lane == src ? inv : lik;
  }
#pragma unroll
  for (int kk = 0; kk < TM; kk++) { const int j = t0 + kk; if (own && j <= lane) A[tri_c0(j, N) + lane - j] = a[kk]; }
  DMC_WSYNC();
}
template <typename T, int LPE, int N, int TM>
DMC_FN void chol_solve_trees(DMC_LDS T* x, const DMC_LDS T* Lm, const DMC_LDS T* b, int lane, int t0, int t1) {
  static_assert(LPE == 64 && N <= LPE, "one lane per unknown, one environment per wave");
  const int i = lane;
  const bool own = i < N;
  const int ci = tri_c0(own ? i : 0, N);
  T row[TM], col[TM], dk[TM];
  const T dinv = own ? Lm[ci] : (T)0;      // 1 / L[i][i]
#pragma unroll
  for (int kk = 0; kk < TM; kk++) {
    const int k = t0 + kk;
    row[kk] = (own && k < i) ? Lm[tri_c0(k, N) + i - k] : (T)0;
    col[kk] = (own && k > i && k < t1) ? Lm[ci + k - i] : (T)0;
    dk[kk] = lane_read(dinv, k);
  }
  T sreg = own ? b[i] : (T)0;
#pragma unroll
  for (int kk = 0; kk < TM; kk++) {
    const int k = t0 + kk;
    const T xk = lane_read(sreg, k) * dk[kk];
    if (i == k) sreg = xk;
    if (i > k && own) sreg -= row[kk]*xk;
  }
#pragma unroll
  for (int kk = TM - 1; kk >= 0; kk--) {
    const int k = t0 + kk;
    const T xk = lane_read(sreg, k) * dk[kk];
    if (i == k) sreg = xk;
    if (i < k && k < t1 && own) sreg -= col[kk]*xk;
  }
  if (own) x[i] = sreg;
  DMC_WSYNC();
}
#endif
// x = (L L')^-1 b (x may alias b); Lm as produced by chol_factor_lds
//   n <= LPE : lane i carries x[i] in a register; the pivot value travels by a
//              cross-lane broadcast, no LDS round trip, no fence inside the loops.
template <typename T, int LPE>
DMC_FN void chol_solve_lds(DMC_LDS T* x, const DMC_LDS T* Lm, const DMC_LDS T* b, int n, int lane) {
  if (n <= LPE && LPE > 1) {
    const int i = lane;
    const int ci = tri_c0(i < n ? i : 0, n);
    T sreg = i < n ? b[i] : (T)0;
    for (int k = 0; k < n; k++) {
      const int ck = tri_c0(k, n);
      const T lik = (i > k && i < n) ? Lm[ck + i - k] : (T)0;
      const T xk = wave_bcast<LPE>(sreg, k) * Lm[ck];
      if (i == k) sreg = xk;
      if (i > k && i < n) sreg -= lik*xk;
    }
    for (int k = n - 1; k >= 0; k--) {
      const T lki = i < k ? Lm[ci + k - i] : (T)0;
      const T xk = wave_bcast<LPE>(sreg, k) * Lm[tri_c0(k, n)];
      if (i == k) sreg = xk;
      if (i < k) sreg -= lki*xk;
    }
    if (i < n) x[i] = sreg;
    DMC_WSYNC();
    return;
  }
  for (int i = lane; i < n; i += LPE) x[i] = b[i];
  DMC_WSYNC();
  for (int k = 0; k < n; k++) {
    const int ck = tri_c0(k, n);
    const T xk = x[k] * Lm[ck];
    DMC_WSYNC();
    if (lane == 0) x[k] = xk;
    for (int i = k + 1 + lane; i < n; i += LPE) x[i] -= Lm[ck + i - k]*xk;
    DMC_WSYNC();
  }
  for (int k = n - 1; k >= 0; k--) {
    const T xk = x[k] * Lm[tri_c0(k, n)];
    DMC_WSYNC();
    if (lane == 0) x[k] = xk;
    for (int i = lane; i < k; i += LPE) x[i] -= Lm[tri_c0(i, n) + k - i]*xk;
    DMC_WSYNC();
  }
}
// line-search evaluation: cost / derivatives of the piecewise quadratic at alpha
template <typename T> struct LSPoint { T alpha, cost, d0, d1; };
// fp32: the line search works on the cost RELATIVE to alpha = 0.  Its points are only ever compared with each other,
// and the total cost of a stiff contact configuration is ~1e6 -- one fp32 ulp of that is 0.06, more than the
// improvement of a late Newton iteration: with absolute costs the search answered "no improvement" (alpha = 0) two
// iterations before the fp64 solver converged, leaving one-step errors of 1e-4 on the 62-dof model.  The relative cost
// is also what the solver's stopping test uses as the iteration's improvement (exact to ~1e-6 of itself instead of to
// an ulp of the total).  fp64 keeps the oracle's absolute form, operation for operation.
template <typename T> DMC_DEV constexpr bool ls_relative() { return sizeof(T) == 4; }
// fp32, round 5: the relative form ANCHORED on the gradient.  The derivative of the cost along the search direction at
// alpha = 0 is grad . search -- a sum of nv small products of a vector the solver already holds to ~8 ulp of |M a| --
// but the line search rebuilt it as s.(M a - qfrc_smooth) + sum_rows D jar jv: two sums of magnitude 1e2 - 1e3 that
// cancel to 1e-4 in a late iteration, i.e. to the rounding noise of their terms.  The search then saw "no descent"
// (alpha = 0) with the gradient still 160 x above its floor and the solve stopped two iterations before the fp64 one:
// the residual qacc error of 3e-4 |qacc| WAS the one-step fp32 error of the 62-dof model (dm_control_amd/DESIGN.md
// section 2, measured on the host build: 8.9e-6 -> see there).  Anchored: the linear coefficient is grad . search, and
// every row contributes only what CHANGES against its zone at alpha = 0 (a row in the same zone at both ends adds
// nothing to it, exactly; a switching row adds its own small term).  Same function of alpha in exact arithmetic.
// Where: measured on the device (profiles/r05_ab_variants.log, one box): the 62-dof walker (general rows, cones in the middle
// zone) -- one-step error p99 1.9e-6 -> 4.2e-7, max 3.9e-5 -> 8.8e-6, for 1.1 % of its speed; the 27-dof humanoid (one-sided
// quadratic rows: the anchored form is a few instructions there) max 2.3e-6 -> 8.2e-7, free; the 30-dof soccer model
// (general rows; already at 2.5e-7 without it) paid 5.5 % for nothing.  So: every model whose rows are all one-sided
// quadratic, and the general-row models with more than 32 dofs (StepCore::anchored()).
template <typename T> DMC_DEV constexpr bool ls_anchored() { return ls_relative<T>(); }
// Returned as a 4-vector {alpha, cost, d0, d1}: a pointer argument pins the caller's
// points in scratch memory, and returning the struct itself by value measured 2x
// slower on the whole kernel (MI355X, ROCm 7.2) -- the vector comes back in v0..v3.
#ifdef DMC_HOST_EMU
template <typename T> struct LSVec { T v[4]; DMC_DEV T operator[](int i) const { return v[i]; } };
#define DMC_LSVEC(T) LSVec<T>
#else
#define DMC_LSVEC(T) T __attribute__((ext_vector_type(4)))
#endif
template <typename T, int LPE>
DMC_FN DMC_LSVEC(T) ls_eval_lds(T a, const DMC_LDS T* jar_, const DMC_LDS T* jv_, const DMC_LDS T* D_,
                                T qg0, T qg1, T qg2, int nefc, int lane) {
  LSPoint<T> pt;
  LSPoint<T>* p = &pt;
  p->alpha = a;
  T q0 = 0, q1 = 0, q2 = 0;
  for (int i = lane; i < nefc; i += LPE) {
    const T jar = jar_[i], jv = jv_[i];
    if (ls_relative<T>()) {
      // cost RELATIVE to alpha = 0 (see ls_relative): the constant 1/2 D jar^2 of a row active at both ends drops out,
      // a row that switches between 0 and alpha contributes it with the sign of the switch
      const bool act_a = jar + a*jv < 0, act_0 = jar < 0;
      if (act_a | act_0) {
        const T D = D_[i], dj0 = D*jar;
        if (act_a) { if (!ls_anchored<T>()) q1 += jv*dj0; q2 += (T)0.5*D*jv*jv; }
        if (act_a != act_0) { q0 += (act_a ? (T)0.5 : (T)-0.5)*jar*dj0; if (ls_anchored<T>()) q1 += act_a ? jv*dj0 : -(jv*dj0); }
      }
    } else if (jar + a*jv < 0) {
      const T D = D_[i], dj0 = D*jar;
      q0 += (T)0.5*jar*dj0; q1 += jv*dj0; q2 += (T)0.5*D*jv*jv;
    }
  }
  q0 = group_sum<LPE>(q0) + qg0; q1 = group_sum<LPE>(q1) + qg1; q2 = group_sum<LPE>(q2) + qg2;
  p->cost = a*a*q2 + a*q1 + q0;
  p->d0 = 2*a*q2 + q1;
  p->d1 = 2*q2;
  if (p->d1 <= 0) p->d1 = (T)DMC_MINVAL;
  DMC_LSVEC(T) rv = {pt.alpha, pt.cost, pt.d0, pt.d1};
  return rv;
}

// ---------------------------------------------------------------------------
// the step
// ---------------------------------------------------------------------------
// where a StepCore finds its layout: a runtime struct (generic kernel) or a
// build-time constant (model-specialised kernels, see step_kernel.hip.h)
struct DynLayoutSrc {
  static constexpr int kNV = 0;   // nv only known at run time
  static constexpr int kJGlobal = -1;   // so is StepDims::jglobal
  static constexpr int kNKin = 0;       // and the size of the kinematic stash
  static constexpr int kTreeMax = 0;    // and StepDims::treemax (the side-by-side tree factorisations are register routines)
  static constexpr int kTreeUni = 0;
  const StepLayout* p;
  DMC_DEV const StepLayout& get() const { return *p; }
};

#ifndef DMC_LS_SLOPE_ULPS
#define DMC_LS_SLOPE_ULPS 16   // fp32 line search: slope floor in ulp of the slope at alpha = 0 (primal_search)
#endif
#ifndef DMC_PRIO_ITER
#define DMC_PRIO_ITER 2   // Newton iteration from which a wave raises its issue priority (fwd_constraint)
#endif
template <typename T, int LPE, typename LS> struct StageFns;   // out-of-line stage entry points (below)

// What a task epilogue (StepIO::task_args; suite/fused_env.py) may read straight from the environment's LDS scratch at the
// end of its launch instead of reading the launch's stores back from global memory: the arrays store_state /
// store_outputs copy row for row.  (Not here: site poses, which only exist as stores; xquat / geom_xmat, which the
// large models keep in global memory.)
template <typename T>
struct TaskLds { const T *qpos, *qvel, *ctrl, *act, *sensordata, *xpos, *xmat, *xipos, *subtree_com, *geom_xpos, *cvel; };

template <typename T, int LPE, typename LS = DynLayoutSrc>
struct StepCore {
  LS ls;
  const StepLayout& L;
  const StepOpts<T>& o;
  const int* mi;
  const T* mr;
  const int* gc;          // cold int tables, global memory (STEP_MODEL_COLD_TABLES)
  T* s;
  int* si;
  int lane;
  double time_;           // simulation time of this env (group-uniform)
  bool hsplit;            // this solve's H is block diagonal over the kinematic trees (h_split; group-uniform)


  DMC_DEV StepCore(LS ls_, const StepOpts<T>& o_, const int* mi_, const T* mr_, const int* gc_, T* s_, int* si_, int lane_)
      : ls(ls_), L(ls_.get()), o(o_), mi(mi_), mr(mr_),
        // small models stage the cold tables in LDS right behind the real tables (step_kernel.hip.h): derive the
        // pointer from `mr` so that the loads compile to LDS reads
#ifndef DMC_HOST_EMU
        gc(ls_.get().d.coldlds ? reinterpret_cast<const int*>(mr_ + ls_.get().n_mr_lds) : gc_),
#else
        gc(gc_),
#endif
        s(s_), si(si_), lane(lane_), time_(0), hsplit(false) {}

#define MI(n) (mi + L.mi_##n)
#define MR(n) (mr + L.mr_##n)
#define MRC(n) (mrc() + L.mr_##n)   // a cold real table (STEP_MODEL_COLD_REAL_TABLES)
#define MRS(n) ((L.d.sitegl ? (const T*)o.g_mr : (const T*)mr) + L.mr_##n)   // a site table (STEP_MODEL_SITE_REAL_TABLES)
#define GC(n) (gc + L.mc_##n)
#define S(n) (s + L.s_##n)
#define SG(n) (sg_##n())      // arrays that live in global scratch at offload level 3
#define SI(n) (si + L.si_##n)
#define FOR_LANES(i, n) for (int i = lane; i < (n); i += LPE)

  // Cold real tables: in LDS with the others, or (large models) in global memory.  The model-specialised kernels know
  // which at compile time; the round trip through the global address space lets the compiler emit global_load.
  DMC_DEV const T* mrc() const {
#ifndef DMC_HOST_EMU
    if constexpr (LS::kJGlobal >= 2) return (const T*)(const DMC_GLB T*)o.g_mr;
    else if constexpr (LS::kJGlobal >= 0) return mr;
    else
#endif
    return L.d.jglobal >= 2 ? (const T*)o.g_mr : mr;
  }
  // Opaque copy of an env index: stops the compiler from forming HBM addresses long before
  // they are used and carrying them across the out-of-line stage calls (callee-saved VGPRs,
  // spilled to scratch once there are too many).
  // the stash epoch of the launch, kept in a spare slot of the env's `misc` reals (bit-cast: every int32 survives a float / double slot)
  DMC_DEV void set_epoch(int e) { if (sizeof(T) == 4) { float f; __builtin_memcpy(&f, &e, 4); S(misc)[MISC_EPOCH] = (T)f; } else S(misc)[MISC_EPOCH] = (T)e; }
  DMC_DEV int get_epoch() const { if (sizeof(T) == 4) { const float f = (float)S(misc)[MISC_EPOCH]; int e; __builtin_memcpy(&e, &f, 4); return e; } return (int)S(misc)[MISC_EPOCH]; }
  DMC_DEV static int late(int env) {
#ifndef DMC_HOST_EMU
    asm volatile("" : "+v"(env));
#endif
    return env;
  }
  // StepIO's arrays are global memory: said so, their accesses are global_load / global_store instead of flat ones (a flat
  // access counts on lgkmcnt as well, so every LDS read after a flat store waited for the store's address to resolve)
  template <typename P> DMC_DEV static DMC_GLB P* G(P* p) { return (DMC_GLB P*)p; }
  template <typename P> DMC_DEV static const DMC_GLB P* G(const P* p) { return (const DMC_GLB P*)p; }
  // ---- state I/O (SoA in HBM <-> LDS) --------------------------------------
  // the per-env stash: everything the stages keep in LDS (persistent reals + all ints), env-major in HBM
  DMC_DEV bool load_stash(const StepIO<T>& io, int env) {
    env = late(env);
    const int* hi = io.stash_i + (size_t)env*(L.n_si + 4);
    if (hi[0] != get_epoch()) return false;      // group-uniform: never written, or written before the last host edit
    const T* hr = io.stash_r + (size_t)env*L.n_keep;
    FOR_LANES(i, L.n_keep) s[i] = hr[i];
    FOR_LANES(i, L.n_si) si[i] = hi[4 + i];
    DMC_WSYNC();
    return true;
  }
  DMC_DEV void store_stash(const StepIO<T>& io, int env, bool valid) {
    env = late(env);
    int* hi = io.stash_i + (size_t)env*(L.n_si + 4);
    if (valid) {
      T* hr = io.stash_r + (size_t)env*L.n_keep;
      FOR_LANES(i, L.n_keep) hr[i] = s[i];
      FOR_LANES(i, L.n_si) hi[4 + i] = si[i];
    }
    if (lane == 0) hi[0] = valid ? get_epoch() : 0;      // the epoch the launch STARTED in: a bump that lands mid-launch must not be adopted
  }
  // ---- hand-off between the pieces of a sliced item (StepIO::slices / hand) ---------------------------------------------
  // The record travels between waves that may sit on different CUs.  Written with relaxed agent-scope 8-byte atomic stores
  // (global_store_dwordx2 sc1: write-through), read with relaxed agent-scope 8-byte atomic loads (sc1: not served by the
  // reader's L1, which another CU's stores never refresh); the producer waits for vmcnt(0) before it raises the item's
  // flag and the consumer polls the flag before it loads (step_kernel_body).
#ifndef DMC_HOST_EMU
  typedef unsigned long long u64h;
  DMC_DEV static int hand_words(int n) { return sizeof(T) == 8 ? n : (n + 1) / 2; }
  DMC_DEV static int hand_record_words(const StepDims& d) { return hand_words(d.nq) + 2*hand_words(d.nv) + hand_words(d.na) + 1; }
  DMC_DEV void hand_put(u64h* dst, const T* src, int n) {
    if (sizeof(T) == 8) { FOR_LANES(i, n) { u64h w; const T v = src[i]; __builtin_memcpy(&w, &v, 8); __hip_atomic_store(dst + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
    else FOR_LANES(i, (n + 1) / 2) {
      const float lo = (float)src[2*i], hi = 2*i + 1 < n ? (float)src[2*i + 1] : 0.0f;
      const u64h w = (u64h)__float_as_uint(lo) | (u64h)__float_as_uint(hi) << 32;
      __hip_atomic_store(dst + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  DMC_DEV void hand_get(T* dst, const u64h* src, int n) {
    if (sizeof(T) == 8) { FOR_LANES(i, n) { const u64h w = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); T v; __builtin_memcpy(&v, &w, 8); dst[i] = v; } }
    else FOR_LANES(i, (n + 1) / 2) {
      const u64h w = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dst[2*i] = (T)__uint_as_float((unsigned)w);
      if (2*i + 1 < n) dst[2*i + 1] = (T)__uint_as_float((unsigned)(w >> 32));
    }
  }
  DMC_DEV void store_handoff(const StepIO<T>& io, int env) {
    env = late(env);
    const int nq = L.d.nq, nv = L.d.nv, na = L.d.na;
    u64h* h = io.hand + (size_t)env * io.hand_n;
    hand_put(h, S(qpos), nq); h += hand_words(nq);
    hand_put(h, S(qvel), nv); h += hand_words(nv);
    hand_put(h, S(qacc_warmstart), nv); h += hand_words(nv);
    if (na) { hand_put(h, S(act), na); h += hand_words(na); }
    if (lane == 0) { u64h w; const double t = time_; __builtin_memcpy(&w, &t, 8); __hip_atomic_store(h, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    const int B = io.B;
    // this piece's warnings (a later piece on another CU adds its own: where atomics execute, not through an L1)
    if (LPE >= 16) { if (lane < DMC_NWARNING) { const int w = SI(imisc)[IM_WARN + lane]; if (w) atomicAdd(io.warning + (size_t)lane*B + env, w); } }
    else if (lane == 0) for (int k = 0; k < DMC_NWARNING; k++) if (SI(imisc)[IM_WARN + k]) atomicAdd(io.warning + (size_t)k*B + env, SI(imisc)[IM_WARN + k]);
    // ctrl zeroed by a BADCTRL warning (mj_fwdActuation) stays zeroed for the pieces that follow
    if (SI(imisc)[IM_WARN + DMC_WARN_BADCTRL]) FOR_LANES(i, L.d.nu) {
      if (sizeof(T) == 8) { u64h w; const T v = S(ctrl)[i]; __builtin_memcpy(&w, &v, 8); __hip_atomic_store((u64h*)(io.ctrl + (size_t)i*B + env), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else __hip_atomic_store((unsigned*)(io.ctrl + (size_t)i*B + env), __float_as_uint((float)S(ctrl)[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  DMC_DEV void load_handoff(const StepIO<T>& io, int env) {
    env = late(env);
    const int nq = L.d.nq, nv = L.d.nv, na = L.d.na, B = io.B;
    const u64h* h = io.hand + (size_t)env * io.hand_n;
    hand_get(S(qpos), h, nq); h += hand_words(nq);
    hand_get(S(qvel), h, nv); h += hand_words(nv);
    hand_get(S(qacc_warmstart), h, nv); h += hand_words(nv);
    if (na) { hand_get(S(act), h, na); h += hand_words(na); FOR_LANES(i, na) S(act_dot)[i] = 0; }
    { const u64h w = __hip_atomic_load(h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); double t; __builtin_memcpy(&t, &w, 8); time_ = t; }
    FOR_LANES(i, nv) S(qfrc_applied)[i] = io.qfrc_applied ? io.qfrc_applied[(size_t)i*B + env] : (T)0;      // (an input: constant over the launch)
    FOR_LANES(i, L.d.nu) {      // (an earlier piece may have zeroed it: BADCTRL)
      if (sizeof(T) == 8) { const u64h w = __hip_atomic_load((const u64h*)(io.ctrl + (size_t)i*B + env), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); T v; __builtin_memcpy(&v, &w, 8); S(ctrl)[i] = v; }
      else S(ctrl)[i] = (T)__uint_as_float(__hip_atomic_load((const unsigned*)(io.ctrl + (size_t)i*B + env), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
#else
  DMC_DEV void store_handoff(const StepIO<T>&, int) {}
  DMC_DEV void load_handoff(const StepIO<T>&, int) {}
#endif
  // ---- kinematic stash ------------------------------------------------------------------------------------------
  // A legacy Physics.step() ends with mj_step1 at the new state and the next one begins with mj_step2 on those
  // results (engine.py:147-162).  Across launches LDS is lost, so the opening position / velocity stage is recomputed
  // -- except for what depends on (qpos, qvel) ALONE: body / geom poses, the COM frame (subtree_com, cinert, cdof)
  // and the velocities (cvel, cdof_dot), one contiguous range of the scratch.  The trailing stage leaves them in
  // global memory together with the (qpos, qvel) they belong to; the next launch takes them back iff its state is
  // bit-equal (and no model constant was edited: the epoch), so writes to bound tensors need no notification.
  DMC_DEV int kin_count() const { return L.s_qM - L.s_xpos; }
  DMC_DEV bool load_kstash(const StepIO<T>& io, int env) {
    env = late(env);
    if (io.kstash_i[env] != get_epoch()) return false;
    const int nq = L.d.nq, nv = L.d.nv, nk = kin_count();
    const T* h = io.kstash + (size_t)env*(nq + nv + nk);
    int bad = 0;
    FOR_LANES(i, nq) if (!(h[i] == S(qpos)[i])) bad = 1;
    FOR_LANES(i, nv) if (!(h[nq + i] == S(qvel)[i])) bad = 1;
    if (group_max<LPE>(bad)) return false;
    FOR_LANES(i, nk) S(xpos)[i] = h[nq + nv + i];
    DMC_WSYNC();
    kstash_env_geoms();
    return true;
  }
  DMC_DEV void store_kstash(const StepIO<T>& io, int env) {
    env = late(env);
    const int nq = L.d.nq, nv = L.d.nv, nk = kin_count();
    DMC_GLB T* h = G(io.kstash) + (size_t)env*(nq + nv + nk);
    FOR_LANES(i, nq) h[i] = S(qpos)[i];
    FOR_LANES(i, nv) h[nq + i] = S(qvel)[i];
    FOR_LANES(i, nk) h[nq + nv + i] = S(xpos)[i];
    if (lane == 0) G(io.kstash_i)[env] = get_epoch();
  }
  // ---- launch-entry loads ------------------------------------------------------------------------------------------
  // Everything a launch reads from HBM before it can start -- the env's launch override, its state, the tag of its
  // kinematic stash, the (qpos, qvel) the stash belongs to and the stash itself -- is requested in ONE batch of
  // independent loads (issued by the kernel before it stages the model tables, so that one HBM round trip covers
  // tables, state and stash) and written to the env's LDS scratch, which nobody else touches; what is left for run()
  // is a few flags.  Loading them where they are consumed cost four dependent round trips at the head of every launch
  // (override -> state -> tag -> compare -> data).  The stash is copied SPECULATIVELY: when it turns out stale the
  // kinematics pass overwrites it.  Lane i carries element i of every field: models with more than LPE coordinates,
  // and launches that use the full stash (whose copy of the state would overwrite this one), keep the in-place loads.
  // Launch features only some callers need -- the substep probe, legacy_step 2 (step + mj_forward), the implicitfast
  // integrator -- are compiled into the generic kernels and into the model-specialised kernels of the units that define
  // DMC_STATIC_FEATURES 1 (the large models, fp64).  The fp32 kernels specialised for the small suite models leave them
  // out (a launch that needs one runs the generic kernel: launch_step_t): on the 9-dof model they cost 6 VGPRs, 14
  // spilled SGPRs, 2.6 KB of code and 1.2 % of the launch (A/B on one box, profiles/r04_ab_vs_round3.log).
  static constexpr bool kFeat = LS::kNV <= 0 || DMC_STATIC_FEATURES;
  // sliced items (StepIO::slices): compiled into the generic kernels and the kernels specialised for models of more than
  // 16 dofs -- the ones whose queued multi-step launches end on a long item; the small models' kernels sit at their
  // register budget (the hand-off code cost the queued 9-dof kernel 460 B of scratch per lane) and their items are short
  static constexpr bool kSlices = LS::kNV <= 0 || LS::kNV > 16;
  static constexpr int kPFKin = LS::kNKin > 0 ? ((LS::kNKin + LPE - 1) / LPE < 24 ? (LS::kNKin + LPE - 1) / LPE : 24) : 0;
  struct Entry { int em, fast, kvalid, epoch; };
  struct EntryRegs {
    T qpos, qvel, warm, qfrc, ctrl, act, kq, kv;
    int ktag, epoch;
    T kd[kPFKin ? kPFKin : 1];
  };
  DMC_DEV static bool kstash_applies(const StepOpts<T>& o, const StepIO<T>& io, int mode, int legacy) {
    return io.kstash && mode == 0 && legacy && o.integrator != DMC_INT_RK4 && !io.stash_r;
  }
  DMC_DEV static void entry_issue(const StepLayout& L, const StepOpts<T>& o, const StepIO<T>& io, int env, int lane, int mode,
                                  int legacy, Entry* e, EntryRegs* r) {
    e->em = io.env_mode ? io.env_mode[env] : 0;
    e->kvalid = 0;
    e->epoch = *io.epoch;      // read ONCE per launch: the tags this launch writes carry the epoch it started in
#ifdef DMC_HOST_EMU
    e->fast = 0;
#else
    e->fast = (L.d.nq <= LPE && L.d.nv <= LPE && L.d.nu <= LPE && L.d.na <= LPE && !io.stash_r) ? 1 : 0;
#endif
    r->ktag = 0; r->epoch = 1;
    if (!e->fast) return;
    const size_t B = (size_t)io.B;
    const int nq = L.d.nq, nv = L.d.nv;
    r->qpos = lane < nq ? G(io.qpos)[lane*B + env] : (T)0;
    r->qvel = lane < nv ? G(io.qvel)[lane*B + env] : (T)0;
    r->warm = lane < nv ? G(io.qacc_warmstart)[lane*B + env] : (T)0;
    r->qfrc = (io.qfrc_applied && lane < nv) ? G(io.qfrc_applied)[lane*B + env] : (T)0;
    r->ctrl = lane < L.d.nu ? G(io.ctrl)[lane*B + env] : (T)0;
    r->act = (L.d.na && lane < L.d.na) ? G(io.act)[lane*B + env] : (T)0;
    if (!kstash_applies(o, io, mode, legacy)) return;
    r->ktag = G(io.kstash_i)[env]; r->epoch = e->epoch;
    const int nk = L.s_qM - L.s_xpos;
    const DMC_GLB T* h = G(io.kstash) + (size_t)env*(nq + nv + nk);
    r->kq = lane < nq ? h[lane] : (T)0;
    r->kv = lane < nv ? h[nq + lane] : (T)0;
#pragma unroll
    for (int k = 0; k < kPFKin; k++) { const int i = lane + k*LPE; r->kd[k] = i < nk ? h[nq + nv + i] : (T)0; }
  }
  // s: the LDS scratch of the env (of this lane's group)
  DMC_DEV static void entry_commit(const StepLayout& L, const StepIO<T>& io, int env, int lane, Entry* e, const EntryRegs& r, T* s) {
    if (!e->fast) return;
    const int nq = L.d.nq, nv = L.d.nv;
    if (lane < nq) s[L.s_qpos + lane] = r.qpos;
    if (lane < nv) { s[L.s_qvel + lane] = r.qvel; s[L.s_qacc_warmstart + lane] = r.warm; s[L.s_qfrc_applied + lane] = r.qfrc; }
    if (lane < L.d.nu) s[L.s_ctrl + lane] = r.ctrl;
    if (L.d.na && lane < L.d.na) s[L.s_act + lane] = r.act;
    if (r.ktag != r.epoch) return;
    const int bad = ((lane < nq && !(r.kq == r.qpos)) || (lane < nv && !(r.kv == r.qvel))) ? 1 : 0;
    if (group_max<LPE>(bad)) return;
    const int nk = L.s_qM - L.s_xpos;
#pragma unroll
    for (int k = 0; k < kPFKin; k++) { const int i = lane + k*LPE; if (i < nk) s[L.s_xpos + i] = r.kd[k]; }
    if (kPFKin*LPE < nk) {
      const T* h = io.kstash + (size_t)env*(nq + nv + nk) + nq + nv;
      for (int i = lane + kPFKin*LPE; i < nk; i += LPE) s[L.s_xpos + i] = h[i];
    }
    e->kvalid = 1;
  }
  // per-environment geoms are inputs the stash does not compare: their world poses are taken from the rows again
  DMC_DEV void kstash_env_geoms() {
    if (!o.eg_n) return;
    FOR_LANES(g, L.d.ngeom) {
      const int k = o.eg_slot[g];
      if (k < 0) continue;
      const T* eg = (const T*)o.eg_data + (size_t)16*k*o.eg_B + SI(imisc)[IM_ENV];
      for (int j = 0; j < 3; j++) S(geom_xpos)[3*g + j] = eg[(size_t)j*o.eg_B];
      for (int j = 0; j < 9; j++) SG(geom_xmat)[9*g + j] = eg[(size_t)(3 + j)*o.eg_B];
    }
    DMC_WSYNC();
  }
  DMC_DEV void load_state(const StepIO<T>& io, int env, bool have_stash, const Entry& en, bool from_hand = false) {
    const int B = io.B;
    if (kSlices && from_hand) load_handoff(io, env);      // a later piece of a sliced item: the state the previous piece handed over
    else if (en.fast) {      // the state is in LDS already (entry_commit)
      if (L.d.na && lane < L.d.na) S(act_dot)[lane] = 0;
      time_ = io.time[env];      // (not needed before the state is stored: nothing waits for it here)
    } else {
      FOR_LANES(i, L.d.nq) S(qpos)[i] = io.qpos[(size_t)i*B + env];
      FOR_LANES(i, L.d.nv) {
        S(qvel)[i] = io.qvel[(size_t)i*B + env];
        S(qacc_warmstart)[i] = io.qacc_warmstart[(size_t)i*B + env];
        S(qfrc_applied)[i] = io.qfrc_applied ? io.qfrc_applied[(size_t)i*B + env] : (T)0;
      }
      FOR_LANES(i, L.d.nu) S(ctrl)[i] = io.ctrl[(size_t)i*B + env];
      if (L.d.na) FOR_LANES(i, L.d.na) { S(act)[i] = io.act[(size_t)i*B + env]; S(act_dot)[i] = 0; }
      time_ = io.time[env];
    }
    if (lane == 0) SI(imisc)[IM_ENV] = env;
    if (have_stash) {      // the derived arrays come from the stash; only this launch's warning counters start at zero
      if (LPE >= 16) { if (lane < DMC_NWARNING) SI(imisc)[IM_WARN + lane] = 0; } else if (lane == 0) for (int k = 0; k < DMC_NWARNING; k++) SI(imisc)[IM_WARN + k] = 0;
      DMC_WSYNC();
      return;
    }
    if (lane == 0) {
      for (int k = 0; k < DMC_NWARNING; k++) SI(imisc)[IM_WARN + k] = 0;
      SI(imisc)[IM_NCON] = 0; SI(imisc)[IM_NEFC] = 0; SI(imisc)[IM_ITER] = 0;
      // world body
      T* xp = S(xpos); T* xq = SG(xquat); T* xm = S(xmat); T* xi = S(xipos);
      xp[0] = xp[1] = xp[2] = 0; xi[0] = xi[1] = xi[2] = 0;
      xq[0] = 1; xq[1] = xq[2] = xq[3] = 0;
      for (int k = 0; k < 9; k++) xm[k] = (k % 4 == 0) ? (T)1 : (T)0;
      for (int k = 0; k < 10; k++) SG(cinert)[k] = 0;
      for (int k = 0; k < 6; k++) S(cvel)[k] = 0;
    }
    if (!L.d.msparse) FOR_LANES(i, L.d.nv * L.d.nv) S(qM)[i] = 0;
    FOR_LANES(i, L.d.nsensordata) S(sensordata)[i] = 0;
    DMC_WSYNC();
  }
  DMC_DEV void store_state(const StepIO<T>& io, int env) {
    env = late(env);
    const int B = io.B;
    FOR_LANES(i, L.d.nq) G(io.qpos)[(size_t)i*B + env] = S(qpos)[i];
    FOR_LANES(i, L.d.nv) {
      G(io.qvel)[(size_t)i*B + env] = S(qvel)[i];
      G(io.qacc_warmstart)[(size_t)i*B + env] = S(qacc_warmstart)[i];
    }
    if (L.d.na) FOR_LANES(i, L.d.na) G(io.act)[(size_t)i*B + env] = S(act)[i];
    if (lane == 0) G(io.time)[env] = time_;
    // (one lane per warning counter: lane 0 walking the nine of them was nine dependent LDS round trips at the end of every wave)
    // (atomics: an earlier piece of a sliced item may have added its own from another CU, which this CU's L1 would not show)
#ifndef DMC_HOST_EMU
    if (LPE >= 16) { if (lane < DMC_NWARNING) { const int w = SI(imisc)[IM_WARN + lane]; if (w) atomicAdd(io.warning + (size_t)lane*B + env, w); } }
    else if (lane == 0) for (int k = 0; k < DMC_NWARNING; k++) if (SI(imisc)[IM_WARN + k]) atomicAdd(io.warning + (size_t)k*B + env, SI(imisc)[IM_WARN + k]);
#else
    if (lane == 0) for (int k = 0; k < DMC_NWARNING; k++) if (SI(imisc)[IM_WARN + k]) G(io.warning)[(size_t)k*B + env] += SI(imisc)[IM_WARN + k];
#endif
    // ctrl may have been zeroed by a BADCTRL warning (mj_fwdActuation semantics)
    if (SI(imisc)[IM_WARN + DMC_WARN_BADCTRL]) FOR_LANES(i, L.d.nu) io.ctrl[(size_t)i*B + env] = S(ctrl)[i];
  }
  DMC_DEV void store_outputs(const StepIO<T>& io, int env, int mask) {
    env = late(env);
    const int B = io.B;
    const int nb = L.d.nbody;
    if (mask & OUT_SENSOR) FOR_LANES(i, L.d.nsensordata) G(io.sensordata)[(size_t)i*B + env] = S(sensordata)[i];
    if (mask & OUT_XPOS) FOR_LANES(i, 3*nb) G(io.xpos)[(size_t)i*B + env] = S(xpos)[i];
    if (mask & OUT_XQUAT) FOR_LANES(i, 4*nb) G(io.xquat)[(size_t)i*B + env] = SG(xquat)[i];
    if (mask & OUT_XMAT) FOR_LANES(i, 9*nb) G(io.xmat)[(size_t)i*B + env] = S(xmat)[i];
    if (mask & OUT_XIPOS) FOR_LANES(i, 3*nb) io.xipos[(size_t)i*B + env] = S(xipos)[i];
    if (mask & OUT_SUBTREE_COM) FOR_LANES(i, 3*nb) G(io.subtree_com)[(size_t)i*B + env] = S(subtree_com)[i];
    if (mask & OUT_GEOM) {
      FOR_LANES(i, 3*L.d.ngeom) io.geom_xpos[(size_t)i*B + env] = S(geom_xpos)[i];
      FOR_LANES(i, 9*L.d.ngeom) io.geom_xmat[(size_t)i*B + env] = SG(geom_xmat)[i];
    }
    if (mask & OUT_SITE) FOR_LANES(sid, L.d.nsite) {
      int b = MI(site_bodyid)[sid]; T v[3], q[4], m[9];
      mul_mat_vec3(v, S(xmat) + 9*b, MRS(site_pos) + 3*sid);
      for (int k = 0; k < 3; k++) io.site_xpos[(size_t)(3*sid + k)*B + env] = S(xpos)[3*b + k] + v[k];
      mul_quat(q, SG(xquat) + 4*b, MRS(site_quat) + 4*sid);
      quat2mat(m, q);
      for (int k = 0; k < 9; k++) io.site_xmat[(size_t)(9*sid + k)*B + env] = m[k];
    }
    if (mask & OUT_QACC) FOR_LANES(i, L.d.nv) io.qacc[(size_t)i*B + env] = S(qacc)[i];
    if (mask & OUT_QFRC) FOR_LANES(i, L.d.nv) {
      io.qfrc_bias[(size_t)i*B + env] = S(qfrc_bias)[i];
      io.qfrc_constraint[(size_t)i*B + env] = S(qfrc_constraint)[i];
      io.qfrc_actuator[(size_t)i*B + env] = S(qfrc_actuator)[i];
    }
    if (mask & OUT_ACTUATOR) FOR_LANES(i, L.d.nu) io.actuator_force[(size_t)i*B + env] = S(actuator_force)[i];
    if (lane == 0) {
      G(io.ncon)[env] = SI(imisc)[IM_NCON]; G(io.nefc)[env] = SI(imisc)[IM_NEFC]; G(io.solver_iter)[env] = SI(imisc)[IM_ITER];
    }
    if ((mask & OUT_CONTACT_IDS) && !(mask & OUT_CONTACT)) {
      const int nc = SI(imisc)[IM_NCON];
      FOR_LANES(c, L.d.nconmax) {
        io.contact_geom1[(size_t)c*B + env] = c < nc ? con_g1(c) : -1;
        io.contact_geom2[(size_t)c*B + env] = c < nc ? con_g2(c) : -1;
      }
    }
    if (mask & OUT_CONTACT) {
      const int nc = SI(imisc)[IM_NCON];
      FOR_LANES(c, L.d.nconmax) {
        bool live = c < nc;
        io.contact_geom1[(size_t)c*B + env] = live ? con_g1(c) : -1;
        io.contact_geom2[(size_t)c*B + env] = live ? con_g2(c) : -1;
        io.contact_dist[(size_t)c*B + env] = live ? S(con_dist)[c] : (T)0;
        for (int k = 0; k < 3; k++) io.contact_pos[(size_t)(3*c + k)*B + env] = live ? S(con_pos)[3*c + k] : (T)0;
        for (int k = 0; k < 9; k++) io.contact_frame[(size_t)(9*c + k)*B + env] = live ? conF()[9*c + k] : (T)0;
        T lf[6] = {0, 0, 0, 0, 0, 0};
        if (live) contact_force_local(c, lf);
        for (int k = 0; k < 6; k++) io.contact_force[(size_t)(6*c + k)*B + env] = lf[k];
      }
    }
    if (mask & OUT_CVEL) FOR_LANES(i, 6*nb) io.cvel[(size_t)i*B + env] = S(cvel)[i];
  }
  DMC_DEV void dump_debug(const StepIO<T>& io, int env) {
    env = late(env);
    if (!io.debug || env >= io.ndebug) return;
    FOR_LANES(i, L.n_sr) io.debug[(size_t)i*io.ndebug + env] = s[i];
    FOR_LANES(i, L.n_si) io.debug_i[(size_t)i*io.ndebug + env] = si[i];
    if (L.n_gs) { const T* g = gscr(); FOR_LANES(i, L.n_gs) io.debug[(size_t)(L.n_sr + i)*io.ndebug + env] = g[i]; }
  }

  // ---- kinematics (mj_kinematics) --------------------------------------------
  // Same transforms as MuJoCo, re-associated for a wide machine:
  //   A. every body in parallel: pose RELATIVE TO ITS PARENT from its own joints
  //      (all trig / joint algebra happens here, off the tree's critical path);
  //   B. per tree depth: compose with the parent's world pose (one quaternion
  //      product + one rotation per body);
  //   C. every joint / body / geom in parallel: anchors, axes, inertial and geom frames.
  // The serial chain is nlevel cheap compositions instead of nlevel full joint
  // evaluations; results differ from the body-by-body order only by rounding.
  DMC_DEV void body_local_pose(int i) {
    T p[3], q[4];
    const int jntadr = MI(body_jntadr)[i], jntnum = MI(body_jntnum)[i];
    if (jntnum == 1 && MI(jnt_type)[jntadr] == DMC_JNT_FREE) {
      const int qa = MI(jnt_qposadr)[jntadr];
      for (int k = 0; k < 3; k++) p[k] = S(qpos)[qa + k];
      for (int k = 0; k < 4; k++) q[k] = S(qpos)[qa + 3 + k];
      normalize4(q);
      for (int k = 0; k < 3; k++) { S(xanchor)[3*jntadr + k] = p[k]; S(xaxis)[3*jntadr + k] = MR(jnt_axis)[3*jntadr + k]; }
    } else {
      const int mid = L.d.nmocap ? MI(body_mocapid)[i] : -1;
      if (mid >= 0) {      // a mocap body: its pose is data (mjData.mocap_pos / mocap_quat), not model (mj_kinematics)
        const size_t e = (size_t)SI(imisc)[IM_ENV], MB = (size_t)o.mocap_B;
        for (int k = 0; k < 3; k++) p[k] = ((const T*)o.mocap_pos)[(size_t)(3*mid + k)*MB + e];
        for (int k = 0; k < 4; k++) q[k] = ((const T*)o.mocap_quat)[(size_t)(4*mid + k)*MB + e];
        normalize4(q);
      } else {
        for (int k = 0; k < 3; k++) p[k] = MRC(body_pos)[3*i + k];
        for (int k = 0; k < 4; k++) q[k] = MRC(body_quat)[4*i + k];
      }
      for (int j = jntadr; j < jntadr + jntnum; j++) {
        const int qa = MI(jnt_qposadr)[j], t = MI(jnt_type)[j];
        T R[9], axis[3], anchor[3];
        quat2mat(R, q);
        mul_mat_vec3(axis, R, MR(jnt_axis) + 3*j);
        mul_mat_vec3(anchor, R, MR(jnt_pos) + 3*j);
        anchor[0] += p[0]; anchor[1] += p[1]; anchor[2] += p[2];
        for (int k = 0; k < 3; k++) { S(xanchor)[3*j + k] = anchor[k]; S(xaxis)[3*j + k] = axis[k]; }
        if (t == DMC_JNT_SLIDE) {
          const T d = S(qpos)[qa] - MR(qpos0)[qa];
          p[0] += axis[0]*d; p[1] += axis[1]*d; p[2] += axis[2]*d;
        } else if (t == DMC_JNT_BALL || t == DMC_JNT_HINGE) {
          T qloc[4], vec[3];
          if (t == DMC_JNT_BALL) { for (int k = 0; k < 4; k++) qloc[k] = S(qpos)[qa + k]; normalize4(qloc); }
          else axisangle2quat(qloc, MR(jnt_axis) + 3*j, S(qpos)[qa] - MR(qpos0)[qa]);
          mul_quat(q, q, qloc);
          rot_vec_quat(vec, MR(jnt_pos) + 3*j, q);
          p[0] = anchor[0] - vec[0]; p[1] = anchor[1] - vec[1]; p[2] = anchor[2] - vec[2];
        }
      }
    }
    if (flat_kin()) { T* lp = S(crb) + 10*i; for (int k = 0; k < 3; k++) lp[k] = p[k]; for (int k = 0; k < 4; k++) lp[3 + k] = q[k]; return; }
    for (int k = 0; k < 3; k++) S(xpos)[3*i + k] = p[k];
    for (int k = 0; k < 4; k++) SG(xquat)[4*i + k] = q[k];
  }
  // Poses composed along each body's own chain instead of one fenced pass per tree depth: the local poses are parked
  // in the (still unused) composite-inertia buffer, every body walks up its ancestors (world = local_root o ... o
  // local_body) with independent loads; results differ from the level passes by rounding only.
  DMC_DEV bool flat_kin() const {
    return true;
  }
  DMC_DEV void body_compose_chain(int i) {
    T p[3], q[4], m[9];
    { const T* lp = S(crb) + 10*i; for (int k = 0; k < 3; k++) p[k] = lp[k]; for (int k = 0; k < 4; k++) q[k] = lp[3 + k]; }
    for (int a = MI(body_parentid)[i]; a != 0; a = MI(body_parentid)[a]) {
      const T* lp = S(crb) + 10*a;
      T qa[4] = {lp[3], lp[4], lp[5], lp[6]}, r[3], qq[4];
      rot_vec_quat(r, p, qa);
      for (int k = 0; k < 3; k++) p[k] = lp[k] + r[k];
      mul_quat(qq, qa, q);
      for (int k = 0; k < 4; k++) q[k] = qq[k];
    }
    normalize4(q);
    quat2mat(m, q);
    for (int k = 0; k < 3; k++) S(xpos)[3*i + k] = p[k];
    for (int k = 0; k < 4; k++) SG(xquat)[4*i + k] = q[k];
    for (int k = 0; k < 9; k++) S(xmat)[9*i + k] = m[k];
  }
  DMC_DEV void body_compose(int i) {
    const int pid = MI(body_parentid)[i];
    T p[3], q[4], m[9];
    if (pid == 0) {
      for (int k = 0; k < 3; k++) p[k] = S(xpos)[3*i + k];
      for (int k = 0; k < 4; k++) q[k] = SG(xquat)[4*i + k];
    } else {
      mul_mat_vec3(p, S(xmat) + 9*pid, S(xpos) + 3*i);
      for (int k = 0; k < 3; k++) p[k] += S(xpos)[3*pid + k];
      mul_quat(q, SG(xquat) + 4*pid, SG(xquat) + 4*i);
    }
    normalize4(q);
    quat2mat(m, q);
    for (int k = 0; k < 3; k++) S(xpos)[3*i + k] = p[k];
    for (int k = 0; k < 4; k++) SG(xquat)[4*i + k] = q[k];
    for (int k = 0; k < 9; k++) S(xmat)[9*i + k] = m[k];
  }
  DMC_DEV void kinematics() {
    for (int i = 1 + lane; i < L.d.nbody; i += LPE) body_local_pose(i);
    DMC_WSYNC();
    DMC_PROF(PROF_X4);
    if (flat_kin()) {
      for (int i = 1 + lane; i < L.d.nbody; i += LPE) body_compose_chain(i);
      DMC_WSYNC();
    } else
    for (int lev = 0; lev < L.d.nlevel; lev++) {
      const int a0 = MI(level_adr)[lev], a1 = MI(level_adr)[lev + 1];
      for (int k = a0 + lane; k < a1; k += LPE) body_compose(MI(level_body)[k]);
      DMC_WSYNC();
    }
    DMC_PROF(PROF_X5);
    FOR_LANES(j, L.d.njnt) {
      const int pid = MI(body_parentid)[MI(jnt_bodyid)[j]];
      if (pid != 0) {
        T a[3], ax[3];
        mul_mat_vec3(a, S(xmat) + 9*pid, S(xanchor) + 3*j);
        mul_mat_vec3(ax, S(xmat) + 9*pid, S(xaxis) + 3*j);
        for (int k = 0; k < 3; k++) { S(xanchor)[3*j + k] = a[k] + S(xpos)[3*pid + k]; S(xaxis)[3*j + k] = ax[k]; }
      }
    }
    for (int i = 1 + lane; i < L.d.nbody; i += LPE) {
      T v[3], q[4], m[9];
      mul_mat_vec3(v, S(xmat) + 9*i, MRC(body_ipos) + 3*i);
      for (int k = 0; k < 3; k++) S(xipos)[3*i + k] = S(xpos)[3*i + k] + v[k];
      mul_quat(q, SG(xquat) + 4*i, MRC(body_iquat) + 4*i);
      quat2mat(m, q);
      for (int k = 0; k < 9; k++) S(ximat)[9*i + k] = m[k];
    }
    FOR_LANES(g, L.d.ngeom) {
      const int k = o.eg_n ? o.eg_slot[g] : -1;
      if (k >= 0) {      // a world-fixed geom with a per-environment pose: its world frame IS that pose
        const T* eg = (const T*)o.eg_data + (size_t)16*k*o.eg_B + SI(imisc)[IM_ENV];
        for (int j = 0; j < 3; j++) S(geom_xpos)[3*g + j] = eg[(size_t)j*o.eg_B];
        for (int j = 0; j < 9; j++) SG(geom_xmat)[9*g + j] = eg[(size_t)(3 + j)*o.eg_B];
        continue;
      }
      const int b = MI(geom_bodyid)[g]; T v[3], q[4], m[9];
      mul_mat_vec3(v, S(xmat) + 9*b, MRC(geom_pos) + 3*g);
      for (int k = 0; k < 3; k++) S(geom_xpos)[3*g + k] = S(xpos)[3*b + k] + v[k];
      mul_quat(q, SG(xquat) + 4*b, MRC(geom_quat) + 4*g);
      quat2mat(m, q);
      for (int k = 0; k < 9; k++) SG(geom_xmat)[9*g + k] = m[k];
    }
    DMC_WSYNC();
  }

  // ---- COM frame: subtree_com, cinert, cdof (mj_comPos) -------------------------
  DMC_DEV void subtree_com_body(int b) {
    T acc[3] = {0, 0, 0};
    for (int c = MI(child_adr)[b]; c < MI(child_adr)[b + 1]; c++) {
      const int ch = MI(child_list)[c];
      for (int k = 0; k < 3; k++) acc[k] += S(subtree_usum)[3*ch + k];
    }
    const T mass = MR(body_mass)[b];
    for (int k = 0; k < 3; k++) acc[k] += S(xipos)[3*b + k] * mass;
    for (int k = 0; k < 3; k++) S(subtree_usum)[3*b + k] = acc[k];
    if (MR(body_subtreemass)[b] < (T)DMC_MINVAL) for (int k = 0; k < 3; k++) S(subtree_com)[3*b + k] = S(xipos)[3*b + k];
    else { const T sc = MR(body_invsubtreemass)[b]; for (int k = 0; k < 3; k++) S(subtree_com)[3*b + k] = acc[k] * sc; }
  }
  DMC_DEV void com_pos() {
    if (L.d.dfs) {      // subtree centres of mass in one pass over the id ranges of the subtrees (see crb_mass_matrix)
      FOR_LANES(b, L.d.nbody) {
        const int e = MI(body_subend)[b];
        T acc[3] = {0, 0, 0};
        for (int c = b; c < e; c++) { const T mass = MR(body_mass)[c]; for (int k = 0; k < 3; k++) acc[k] += S(xipos)[3*c + k] * mass; }
        if (MR(body_subtreemass)[b] < (T)DMC_MINVAL) for (int k = 0; k < 3; k++) S(subtree_com)[3*b + k] = S(xipos)[3*b + k];
        else { const T sc = MR(body_invsubtreemass)[b]; for (int k = 0; k < 3; k++) S(subtree_com)[3*b + k] = acc[k] * sc; }
      }
      DMC_WSYNC();
    } else
    {
    for (int lev = L.d.nlevel - 1; lev >= 0; lev--) {
      const int a0 = MI(level_adr)[lev], a1 = MI(level_adr)[lev + 1];
      for (int k = a0 + lane; k < a1; k += LPE) subtree_com_body(MI(level_body)[k]);
      DMC_WSYNC();
    }
    if (lane == 0) subtree_com_body(0);
    DMC_WSYNC();
    }
    for (int i = 1 + lane; i < L.d.nbody; i += LPE) {
      T off[3], ci[10]; const T* rc = S(subtree_com) + 3*MI(body_rootid)[i];
      for (int k = 0; k < 3; k++) off[k] = S(xipos)[3*i + k] - rc[k];
      inert_com(ci, MRC(body_inertia) + 3*i, S(ximat) + 9*i, off, MR(body_mass)[i]);
      for (int k = 0; k < 10; k++) SG(cinert)[10*i + k] = ci[k];
    }
    FOR_LANES(j, L.d.njnt) {
      const int da = 6*MI(jnt_dofadr)[j], bi = MI(jnt_bodyid)[j], t = MI(jnt_type)[j];
      T off[3], axis[3], cr[3]; const T* rc = S(subtree_com) + 3*MI(body_rootid)[bi];
      for (int k = 0; k < 3; k++) off[k] = rc[k] - S(xanchor)[3*j + k];
      T* cd = S(cdof) + da;
      if (t == DMC_JNT_FREE || t == DMC_JNT_BALL) {
        int skip = 0;
        if (t == DMC_JNT_FREE) {
          for (int i = 0; i < 3; i++) for (int k = 0; k < 6; k++) cd[6*i + k] = (k == 3 + i) ? (T)1 : (T)0;
          skip = 3;
        }
        for (int i = 0; i < 3; i++) {
          axis[0] = S(xmat)[9*bi + i]; axis[1] = S(xmat)[9*bi + 3 + i]; axis[2] = S(xmat)[9*bi + 6 + i];
          cross3(cr, axis, off);
          T* c = cd + 6*(i + skip);
          c[0] = axis[0]; c[1] = axis[1]; c[2] = axis[2]; c[3] = cr[0]; c[4] = cr[1]; c[5] = cr[2];
        }
      } else if (t == DMC_JNT_SLIDE) {
        cd[0] = cd[1] = cd[2] = 0;
        for (int k = 0; k < 3; k++) cd[3 + k] = S(xaxis)[3*j + k];
      } else {
        for (int k = 0; k < 3; k++) axis[k] = S(xaxis)[3*j + k];
        cross3(cr, axis, off);
        cd[0] = axis[0]; cd[1] = axis[1]; cd[2] = axis[2]; cd[3] = cr[0]; cd[4] = cr[1]; cd[5] = cr[2];
      }
    }
    DMC_WSYNC();
  }

  // ---- dense Cholesky / solves in LDS (out-of-line: chol_factor_lds / chol_solve_lds) ----
  // `split`: the matrix is block diagonal over the kinematic trees (M always; H when h_split())
  static constexpr bool kSplit = LS::kTreeMax > 0 && LPE == 64 && LS::kNV <= LPE
#ifdef DMC_NO_TREE_SPLIT
                                 && false
#endif
      ;
  // fp32 models of 33 .. 64 dofs on one wave per environment factor on the matrix cores (chol_factor_tiles)
  static constexpr bool kTiles = sizeof(T) == 4 && LPE == 64 && LS::kNV > 32 && LS::kNV <= 64
#if defined(DMC_NO_CHOL_TILES) || defined(DMC_HOST_EMU)
                                 && false
#endif
      ;
  DMC_DEV void chol_factor_inplace(T* A, int n, bool split = false) {
#ifndef DMC_HOST_EMU
    if constexpr (kSplit) {
      if (split) {
        const int t0 = lane < LS::kNV ? MI(dof_tree0)[lane] : 0, t1 = lane < LS::kNV ? MI(dof_tree1)[lane] : 0;
        chol_factor_trees<T, LPE, LS::kNV, LS::kTreeMax>((DMC_LDS T*)A, lane, t0, t1);
        return;
      }
    }
    if constexpr (kTiles) { chol_factor_tiles<LPE, LS::kNV>((DMC_LDS float*)A, lane); return; }
    if constexpr (LS::kNV > 0 && LS::kNV <= LPE) { chol_factor_rows<T, LPE, LS::kNV>((DMC_LDS T*)A, lane); return; }
#endif
    chol_factor_lds<T, LPE>((DMC_LDS T*)A, n, lane);
  }
  DMC_DEV void chol_solve(T* x, const T* Lm, const T* b, int n, bool split = false) {
#ifndef DMC_HOST_EMU
    if constexpr (kSplit) {
      if (split) {
        const int t0 = lane < LS::kNV ? MI(dof_tree0)[lane] : 0, t1 = lane < LS::kNV ? MI(dof_tree1)[lane] : 0;
        chol_solve_trees<T, LPE, LS::kNV, LS::kTreeMax>((DMC_LDS T*)x, (const DMC_LDS T*)Lm, (const DMC_LDS T*)b, lane, t0, t1);
        return;
      }
    }
    if constexpr (LS::kNV > 0 && LS::kNV <= LPE) { chol_solve_rows<T, LPE, LS::kNV>((DMC_LDS T*)x, (const DMC_LDS T*)Lm, (const DMC_LDS T*)b, lane); return; }
#endif
    chol_solve_lds<T, LPE>((DMC_LDS T*)x, (const DMC_LDS T*)Lm, (const DMC_LDS T*)b, n, lane);
  }
  // ---- per-environment global scratch (StepOpts::gscr, StepLayout::gs_*) -------------------------
  DMC_DEV T* gscr() const { return (T*)o.gscr + (size_t)SI(imisc)[IM_ENV] * L.n_gs; }
  DMC_DEV T* gLM() const { return gscr() + L.gs_LM; }

  // ---- CRB mass matrix + factor (mj_crb, mj_factorM) ----------------------------
  DMC_DEV void crb_mass_matrix() {
    const int nv = L.d.nv;
    if (L.d.dfs) {
      // composite inertias as ONE pass of subtree sums: bodies are numbered depth first, so the subtree of b is the id range
      // [b, subend[b]) -- every (body, component) sums its range with independent loads, instead of nlevel dependent
      // child-accumulation passes (each a chain of five LDS reads).  Same terms as the level passes, summed in id order.
      for (int idx = 10 + lane; idx < 10*L.d.nbody; idx += LPE) {
        const int b = idx / 10, comp = idx - 10*b, e = MI(body_subend)[b];
        T v = SG(cinert)[idx];
        for (int c = b + 1; c < e; c++) v += SG(cinert)[10*c + comp];
        S(crb)[idx] = v;
      }
      DMC_WSYNC();
    } else
    {
    for (int i = 10 + lane; i < 10*L.d.nbody; i += LPE) S(crb)[i] = SG(cinert)[i];
    DMC_WSYNC();
    for (int lev = L.d.nlevel - 2; lev >= 0; lev--) {
      const int a0 = MI(level_adr)[lev], cnt = MI(level_adr)[lev + 1] - a0;
      for (int idx = lane; idx < cnt*10; idx += LPE) {
        const int b = MI(level_body)[a0 + idx/10], comp = idx % 10;
        const int c0 = MI(child_adr)[b], c1 = MI(child_adr)[b + 1];
        if (c1 > c0) {
          T v = S(crb)[10*b + comp];
          for (int c = c0; c < c1; c++) v += S(crb)[10*MI(child_list)[c] + comp];
          S(crb)[10*b + comp] = v;
        }
      }
      DMC_WSYNC();
    }
    }
    DMC_PROF(PROF_X1);
    FOR_LANES(i, nv) {
      T buf[6]; mul_inert_vec(buf, S(crb) + 10*MI(dof_bodyid)[i], S(cdof) + 6*i);
      for (int k = 0; k < 6; k++) S(mbuf)[6*i + k] = buf[k];
    }
    DMC_WSYNC();
    FOR_LANES(p, L.d.nM) {
      const int pk = GC(mpair)[p], i = pk & 0xffff, j = pk >> 16;
      T v = dot_n(S(cdof) + 6*j, S(mbuf) + 6*i, 6);
      if (i == j) v += MR(dof_armature)[i];
      if (L.d.msparse) qMs()[p] = v; else { S(qM)[i*nv + j] = v; S(qM)[j*nv + i] = v; }
    }
    DMC_WSYNC();
    DMC_PROF(PROF_X2);
    factor_M(false);
  }
  // contact frames (9 reals per contact): LDS, or the global scratch of a large model
  DMC_DEV T* conF() const {
#ifndef DMC_HOST_EMU
    if constexpr (LS::kJGlobal >= 2) return (T*)(DMC_GLB T*)(gscr() + L.gs_cf);
    else if constexpr (LS::kJGlobal >= 0) return S(con_frame);
    else
#endif
    return L.d.jglobal >= 2 ? gscr() + L.gs_cf : S(con_frame);
  }
  // ---- level-3 arrays (xquat, geom_xmat, cinert, cdof_dot): LDS, or the global scratch of the largest models ------------
#ifndef DMC_HOST_EMU
#define DMC_SG_ACCESSOR(name)                                                                         \
  DMC_DEV T* sg_##name() const {                                                                      \
    if constexpr (LS::kJGlobal >= 3) return (T*)(DMC_GLB T*)(gscr() + L.gs_##name);                  \
    else if constexpr (LS::kJGlobal >= 0) return s + L.s_##name;                                      \
    else return L.d.jglobal >= 3 ? gscr() + L.gs_##name : s + L.s_##name;                             \
  }
#else
#define DMC_SG_ACCESSOR(name) DMC_DEV T* sg_##name() const { return L.d.jglobal >= 3 ? gscr() + L.gs_##name : s + L.s_##name; }
#endif
  DMC_SG_ACCESSOR(xquat) DMC_SG_ACCESSOR(geom_xmat) DMC_SG_ACCESSOR(cinert) DMC_SG_ACCESSOR(cdof_dot)
#undef DMC_SG_ACCESSOR
  // ---- sparse mass matrix ------------------------------------------------------------
  // the nM tree-sparse entries (msparse models): LDS, or the global scratch of a large model
  DMC_DEV T* qMs() const {
#ifndef DMC_HOST_EMU
    if constexpr (LS::kJGlobal >= 2) return (T*)(DMC_GLB T*)(gscr() + L.gs_M);
    else if constexpr (LS::kJGlobal >= 0) return S(qM);
    else
#endif
    return L.d.jglobal >= 2 ? gscr() + L.gs_M : S(qM);
  }
  // qLH <- M (+ diag), packed by columns: zero fill, then the nonzeros (i, j) of the (i, j) list; the list is
  // read from global memory one trip ahead of its use
  DMC_DEV void scatter_M(const T* diag, T diag_scale, T* dst) {
    const int nv = L.d.nv, nM = L.d.nM;
    if (!L.d.msparse) {
      // dense M (small models): every packed entry is written from its (i, j) -- no zero fill, one fence
      FOR_LANES(idx, L.d.ntri) {
        int i, j;
        tri_unrank(idx, nv, &i, &j);
        T v = S(qM)[i*nv + j];
        if (diag && i == j) v += diag_scale*diag[i];
        dst[idx] = v;
      }
      DMC_WSYNC();
      return;
    }
    FOR_LANES(i, L.d.ntri) dst[i] = 0;
    DMC_WSYNC();
    int pk = lane < nM ? GC(mpair)[lane] : 0;
    for (int p = lane; p < nM; p += LPE) {
      const int i = pk & 0xffff, j = pk >> 16;
      if (p + LPE < nM) pk = GC(mpair)[p + LPE];
      T v = L.d.msparse ? qMs()[p] : S(qM)[i*nv + j];
      if (diag && i == j) v += diag_scale*diag[i];
      dst[tri_at(i, j, nv)] = v;
    }
    DMC_WSYNC();
  }
  // qLH <- M (+ timestep * damping on the diagonal), then its Cholesky factor
  // the factor of M lives in qLM where a model runs noslip (it is needed again after H's factor took qLH), else in qLH
  // Large noslip models (jglobal): factored in qLH, where mj_fwdAcceleration still finds it, and copied to the global
  // scratch, from where the noslip pass brings it back once H's factor is done with qLH.
  DMC_DEV T* M_factor() { return S(qLM); }      // aliases qLH when the model does not run noslip or keeps the copy in global memory
  // dense M (+ diag) factored straight from its rows: lane i loads row i of qM into registers and runs
  // chol_factor_rows' elimination there -- no packed copy (scatter_M), no unranking of packed indices, one fence less
#ifndef DMC_HOST_EMU
  template <int N>
  DMC_DEV void factor_dense_rows(const T* diag, T diag_scale, T* dst) {
    const bool own = lane < N;
    const int i = own ? lane : 0;
    T a[N];
#pragma unroll
    for (int j = 0; j < N; j++) a[j] = S(qM)[i*N + j];      // (the whole row, unpredicated: hess_factor_rows)
    if (diag) {
      const T dv = diag_scale*diag[i];
#pragma unroll
      for (int j = 0; j < N; j++) a[j] = (j == lane) ? a[j] + dv : a[j];
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
      T akk = bcast_rows<LPE, N>(a[k], k);
      if (akk < (T)DMC_MINVAL) akk = (T)DMC_MINVAL;
      const T inv = t_rsqrt(akk);
      const T lik = a[k] * inv;
#pragma unroll
      for (int j = k + 1; j < N; j++) { const T ljk = bcast_rows<LPE, N>(lik, j); a[j] = nmsub<(N <= 16)>(a[j], lik, ljk); }
      a[k] = lane == k ? inv : lik;
    }
    store_factor_rows<T, N>((DMC_LDS T*)dst, a, lane);
    DMC_WSYNC();
  }
#endif
  DMC_DEV void factor_M(bool with_damping, const T* damping = nullptr) {      // damping: the diagonal added as timestep * damping[i]
    T* dst = with_damping ? S(qLH) : M_factor();
#ifndef DMC_HOST_EMU
    if constexpr (LS::kNV > 0 && LS::kNV <= 16 && LS::kNV <= LPE) {
      if (!L.d.msparse) {
        factor_dense_rows<LS::kNV>(with_damping ? damping : (const T*)nullptr, o.timestep, dst);
        DMC_PROF(PROF_X3);
        if (!with_damping && L.d.jglobal && L.d.nslip) { DMC_GLB T* g = (DMC_GLB T*)gLM(); FOR_LANES(k, L.d.ntri) g[k] = dst[k]; }
        return;
      }
    }
#endif
    scatter_M(with_damping ? damping : (const T*)nullptr, o.timestep, dst);
    DMC_PROF(PROF_X3);
    chol_factor_inplace(dst, L.d.nv, true);      // M (+ a diagonal) is block diagonal over the trees
    if (!with_damping && L.d.jglobal && L.d.nslip) {
      DMC_GLB T* g = (DMC_GLB T*)gLM();
      FOR_LANES(i, L.d.ntri) g[i] = dst[i];
    }
  }

  // ---- collision (mj_collision over the static candidate pair list) -------------
  DMC_DEV static void make_frame(T* f) {
    normalize3(f);
    if (t_sqrt(dot3(f + 3, f + 3)) < (T)0.5) {
      f[3] = f[4] = f[5] = 0;
      if (f[1] < (T)0.5 && f[1] > (T)-0.5) f[4] = 1; else f[5] = 1;
    }
    T t = dot3(f, f + 3);
    f[3] -= t*f[0]; f[4] -= t*f[1]; f[5] -= t*f[2];
    normalize3(f + 3);
    cross3(f + 6, f, f + 3);
  }
  struct Hit { T dist, pos[3], nrm[3]; };
  DMC_DEV static int plane_sphere(Hit* h, T margin, const T* ppos, const T* nrm, const T* spos, T radius) {
    T dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
    T dist = dot3(dif, nrm) - radius;
    if (dist > margin) return 0;
    h->dist = dist;
    for (int k = 0; k < 3; k++) { h->pos[k] = spos[k] - nrm[k]*(radius + dist*(T)0.5); h->nrm[k] = nrm[k]; }
    return 1;
  }
  DMC_DEV static int sphere_sphere(Hit* h, T margin, const T* p1, T r1, const T* p2, T r2) {
    T dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    T cdist = t_sqrt(dot3(dif, dif));
    T dist = cdist - r1 - r2;
    if (dist > margin) return 0;
    T n[3];
    if (cdist < (T)DMC_MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else { n[0] = dif[0]/cdist; n[1] = dif[1]/cdist; n[2] = dif[2]/cdist; }
    h->dist = dist;
    for (int k = 0; k < 3; k++) { h->pos[k] = p1[k] + n[k]*(r1 + dist*(T)0.5); h->nrm[k] = n[k]; }
    return 1;
  }
  // h[n] = x with every slot index a compile-time constant, so that the hit
  // buffer stays in registers (a dynamically indexed local array lives in scratch)
  struct Hits { Hit s0, s1, s2, s3; };
  DMC_DEV static void sel_hit(Hit& d, const Hit& x, bool c) {
    d.dist = c ? x.dist : d.dist;
    for (int k = 0; k < 3; k++) { d.pos[k] = c ? x.pos[k] : d.pos[k]; d.nrm[k] = c ? x.nrm[k] : d.nrm[k]; }
  }
  DMC_DEV static void put_hit(Hits* h, int n, const Hit& x) {
    sel_hit(h->s0, x, n == 0); sel_hit(h->s1, x, n == 1); sel_hit(h->s2, x, n == 2); sel_hit(h->s3, x, n == 3);
  }
  // ---- ellipsoid pairs: signed distance = max over unit n of the support-function gap F(n),
  // Newton iteration on the unit sphere; a capsule is its swept sphere minimised over the axis
  // parameter (regula falsi on n.axis).  Same iteration, same operation order as the oracle's
  // "ellipsoid pairs" section; everything lives in registers (constant indexing only).
  struct Quadric { T c[3], R[9], s[3]; };
  DMC_DEV static constexpr T ccd_tol() { return sizeof(T) == 8 ? (T)1e-10 : (T)1e-4; }
  DMC_DEV static constexpr int ccd_maxit() { return sizeof(T) == 8 ? 30 : 12; }
  DMC_DEV static T quadric_support(const Quadric& q, const T* n, T* g, T* wh) {
    T w[3];
    for (int k = 0; k < 3; k++) w[k] = q.s[k]*(q.R[k]*n[0] + q.R[3 + k]*n[1] + q.R[6 + k]*n[2]);
    const T wn = t_sqrt(dot3(w, w));
    if (wn < (T)DMC_MINVAL) { g[0] = g[1] = g[2] = 0; wh[0] = wh[1] = wh[2] = 0; return 0; }
    T v[3];
    for (int k = 0; k < 3; k++) { wh[k] = w[k]/wn; v[k] = q.s[k]*wh[k]; }
    mul_mat_vec3(g, q.R, v);
    return wn;
  }
  DMC_DEV static void quadric_curv(const Quadric& q, const T* wh, T wn, const T* t1, const T* t2, T* K) {
    if (wn < (T)DMC_MINVAL) return;
    T y1[3], y2[3];
    for (int k = 0; k < 3; k++) {
      y1[k] = q.s[k]*(q.R[k]*t1[0] + q.R[3 + k]*t1[1] + q.R[6 + k]*t1[2]);
      y2[k] = q.s[k]*(q.R[k]*t2[0] + q.R[3 + k]*t2[1] + q.R[6 + k]*t2[2]);
    }
    const T a1 = dot3(y1, wh), a2 = dot3(y2, wh);
    K[0] += (dot3(y1, y1) - a1*a1)/wn; K[1] += (dot3(y1, y2) - a1*a2)/wn; K[2] += (dot3(y2, y2) - a2*a2)/wn;
  }
  DMC_DEV static T quadric_gap_value(const Quadric& A, const Quadric& B, const T* n) {
    T g[3], wh[3], d[3] = {B.c[0] - A.c[0], B.c[1] - A.c[1], B.c[2] - A.c[2]};
    const T hA = quadric_support(A, n, g, wh), hB = quadric_support(B, n, g, wh);
    return dot3(n, d) - hA - hB;
  }
  DMC_DEV static T quadric_gap(const Quadric& A, const Quadric& B, T* n, T* gA, T* gB) {
    const T d[3] = {B.c[0] - A.c[0], B.c[1] - A.c[1], B.c[2] - A.c[2]};
    T F = 0;
    for (int it = 0; ; it++) {
      T wA[3], wB[3];
      const T hA = quadric_support(A, n, gA, wA), hB = quadric_support(B, n, gB, wB);
      F = dot3(n, d) - hA - hB;
      if (it >= ccd_maxit()) break;
      T f[9] = {n[0], n[1], n[2], 0, 0, 0, 0, 0, 0};
      make_frame(f);
      const T *t1 = f + 3, *t2 = f + 6;
      const T grad[3] = {d[0] - gA[0] - gB[0], d[1] - gA[1] - gB[1], d[2] - gA[2] - gB[2]};
      const T g1 = dot3(t1, grad), g2 = dot3(t2, grad);
      T K[3] = {F, 0, F};
      quadric_curv(A, wA, hA, t1, t2, K); quadric_curv(B, wB, hB, t1, t2, K);
      const T tr = K[0] + K[2];
      T det = K[0]*K[2] - K[1]*K[1];
      const T floor_ = (T)1e-3*(hA + hB) + (T)DMC_MINVAL;
      const T lmin = (T)0.5*(tr - t_sqrt(t_max((T)0, tr*tr - 4*det)));
      if (lmin < floor_) { const T sh = floor_ - lmin; K[0] += sh; K[2] += sh; det = K[0]*K[2] - K[1]*K[1]; }
      T d1 = (K[2]*g1 - K[1]*g2)/det, d2 = (K[0]*g2 - K[1]*g1)/det;
      if (d1*d1 + d2*d2 < ccd_tol()*ccd_tol()) break;
      T nn[3];
      for (int ls = 0; ; ls++) {
        for (int k = 0; k < 3; k++) nn[k] = n[k] + t1[k]*d1 + t2[k]*d2;
        normalize3(nn);
        if (ls >= 8 || quadric_gap_value(A, B, nn) >= F) break;
        d1 *= (T)0.5; d2 *= (T)0.5;
      }
      n[0] = nn[0]; n[1] = nn[1]; n[2] = nn[2];
    }
    return F;
  }
  DMC_DEV static void quadric_init_dir(const Quadric& A, const Quadric& B, T* n) {
    for (int k = 0; k < 3; k++) n[k] = B.c[k] - A.c[k];
    if (dot3(n, n) < (T)DMC_MINVAL*(T)DMC_MINVAL) { n[0] = 1; n[1] = n[2] = 0; }
    normalize3(n);
  }
  DMC_DEV static int quadric_contact(Hit* h, T margin, const Quadric& A, const Quadric& B, T* n) {
    T gA[3], gB[3];
    const T dist = quadric_gap(A, B, n, gA, gB);
    if (dist > margin) return 0;
    h->dist = dist;
    for (int k = 0; k < 3; k++) { h->pos[k] = (T)0.5*((A.c[k] + gA[k]) + (B.c[k] - gB[k])); h->nrm[k] = n[k]; }
    return 1;
  }
  DMC_DEV static Quadric make_quadric(int type, const T* pos, const T* mat, const T* size) {
    Quadric q;
    for (int k = 0; k < 3; k++) { q.c[k] = pos[k]; q.s[k] = type == DMC_GEOM_ELLIPSOID ? size[k] : size[0]; }
    for (int k = 0; k < 9; k++) q.R[k] = mat[k];
    return q;
  }
  // geom 2 is an ellipsoid; geom 1 a plane-less partner (sphere, capsule or ellipsoid)
  DMC_DEV static int ellipsoid_pair(Hit* h, T margin, int t1, const T* p1, const T* m1, const T* s1,
                                    const T* p2, const T* m2, const T* s2) {
    Quadric A = make_quadric(t1 == DMC_GEOM_CAPSULE ? DMC_GEOM_SPHERE : t1, p1, m1, s1);
    const Quadric B = make_quadric(DMC_GEOM_ELLIPSOID, p2, m2, s2);
    T n[3];
    if (t1 != DMC_GEOM_CAPSULE) { quadric_init_dir(A, B, n); return quadric_contact(h, margin, A, B, n); }
    const T u[3] = {m1[2], m1[5], m1[8]};
    const T hl = s1[1];
    T gA[3], gB[3];
    T tlo = -hl, thi = hl, plo, phi = 0, t;
    for (int k = 0; k < 3; k++) A.c[k] = p1[k] + u[k]*tlo;
    quadric_init_dir(A, B, n);
    quadric_gap(A, B, n, gA, gB); plo = dot3(n, u);
    if (plo <= 0) t = tlo;
    else {
      for (int k = 0; k < 3; k++) A.c[k] = p1[k] + u[k]*thi;
      quadric_gap(A, B, n, gA, gB); phi = dot3(n, u);
      if (phi >= 0) t = thi;
      else {
        t = 0;
        int side = 0;
        for (int it = 0; it < 40; it++) {
          t = (tlo*phi - thi*plo)/(phi - plo);
          for (int k = 0; k < 3; k++) A.c[k] = p1[k] + u[k]*t;
          quadric_gap(A, B, n, gA, gB);
          const T pt = dot3(n, u);
          if (t_abs(pt) < ccd_tol() || thi - tlo < ccd_tol()*hl) break;
          if (pt > 0) { tlo = t; plo = pt; if (side == 1) phi *= (T)0.5; side = 1; }
          else { thi = t; phi = pt; if (side == -1) plo *= (T)0.5; side = -1; }
        }
      }
    }
    for (int k = 0; k < 3; k++) A.c[k] = p1[k] + u[k]*t;
    return quadric_contact(h, margin, A, B, n);
  }
  DMC_DEV static int plane_ellipsoid(Hit* h, T margin, const T* p1, const T* nrm, const T* p2, const T* m2, const T* s2) {
    const Quadric q = make_quadric(DMC_GEOM_ELLIPSOID, p2, m2, s2);
    T g[3], wh[3];
    quadric_support(q, nrm, g, wh);
    const T pt[3] = {p2[0] - g[0], p2[1] - g[1], p2[2] - g[2]};
    const T dif[3] = {pt[0] - p1[0], pt[1] - p1[1], pt[2] - p1[2]};
    const T dist = dot3(dif, nrm);
    if (dist > margin) return 0;
    h->dist = dist;
    for (int k = 0; k < 3; k++) { h->pos[k] = pt[k] - nrm[k]*dist*(T)0.5; h->nrm[k] = nrm[k]; }
    return 1;
  }
  // ---- box pairs (sphere-box, capsule-box, box-box): same constructions, same operation order as the
  // oracle's "box pairs" section.  The clipping polygon is a dynamically indexed local array (scratch
  // memory); the code is compiled out of models without such pairs (d.nbox == 0).
  DMC_DEV static int sphere_box_core(Hit* h, T margin, const T* ps, T r, const T* pb, const T* mb, const T* sb) {
    T dif[3] = {ps[0] - pb[0], ps[1] - pb[1], ps[2] - pb[2]}, cl[3], q[3], nb[3] = {0, 0, 0};
    mul_matT_vec3(cl, mb, dif);
    bool outside = false;
    for (int k = 0; k < 3; k++) { q[k] = t_max(-sb[k], t_min(sb[k], cl[k])); if (q[k] != cl[k]) outside = true; }
    T dist;
    if (outside) {
      const T d[3] = {cl[0] - q[0], cl[1] - q[1], cl[2] - q[2]};
      const T dn = t_sqrt(dot3(d, d));
      dist = dn - r;
      if (dist > margin) return 0;
      for (int k = 0; k < 3; k++) nb[k] = d[k]/dn;
    } else {
      const T d0 = sb[0] - t_abs(cl[0]), d1 = sb[1] - t_abs(cl[1]), d2 = sb[2] - t_abs(cl[2]);
      int best = 0; T depth = d0;
      if (d1 < depth) { depth = d1; best = 1; }
      if (d2 < depth) { depth = d2; best = 2; }
      for (int k = 0; k < 3; k++) if (k == best) { nb[k] = cl[k] >= 0 ? (T)1 : (T)-1; q[k] = nb[k]*sb[k]; }
      dist = -depth - r;
    }
    T nw[3], qw[3];
    mul_mat_vec3(nw, mb, nb); mul_mat_vec3(qw, mb, q);
    h->dist = dist;
    for (int k = 0; k < 3; k++) { h->pos[k] = pb[k] + qw[k] + nw[k]*dist*(T)0.5; h->nrm[k] = -nw[k]; }
    return 1;
  }
  DMC_DEV static T seg_box_dd(const T* p0, const T* u, const T* sb, T t, T* deriv) {
    T f = 0, g = 0;
    for (int k = 0; k < 3; k++) {
      const T x = p0[k] + t*u[k], e = x - t_max(-sb[k], t_min(sb[k], x));
      f += e*e; g += 2*e*u[k];
    }
    *deriv = g;
    return f;
  }
  DMC_DEV static int capsule_box(Hits* hs, T margin, const T* p1, const T* m1, const T* s1, const T* p2, const T* m2, const T* s2) {
    const T axw[3] = {m1[2], m1[5], m1[8]}, dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    T p0[3], u[3];
    mul_matT_vec3(p0, m2, dif); mul_matT_vec3(u, m2, axw);
    const T hl = s1[1];
    T lo = -hl, hi = hl, g, t;
    seg_box_dd(p0, u, s2, lo, &g);
    if (g >= 0) t = lo;
    else {
      seg_box_dd(p0, u, s2, hi, &g);
      if (g <= 0) t = hi;
      else {
        for (int it = 0; it < (sizeof(T) == 8 ? 60 : 30); it++) { t = (T)0.5*(lo + hi); seg_box_dd(p0, u, s2, t, &g); if (g > 0) hi = t; else lo = t; }
        t = (T)0.5*(lo + hi);
      }
    }
    int mask = 0;
    T ps[3];
    for (int k = 0; k < 3; k++) ps[k] = p1[k] + axw[k]*t;
    mask |= sphere_box_core(&hs->s0, margin, ps, s1[0], p2, m2, s2);
    const T t2 = t <= 0 ? hl : -hl;
    if (t_abs(t2 - t) > (T)1e-3*hl) {
      for (int k = 0; k < 3; k++) ps[k] = p1[k] + axw[k]*t2;
      Hit x;
      if (sphere_box_core(&x, margin, ps, s1[0], p2, m2, s2)) { put_hit(hs, mask & 1, x); mask = (mask << 1) | 1; }
    }
    return mask;
  }
  DMC_DEV static int box_box(Hits* hs, T margin, const T* pA, const T* RA, const T* sA, const T* pB, const T* RB, const T* sB) {
    const T d[3] = {pB[0] - pA[0], pB[1] - pA[1], pB[2] - pA[2]};
    T colA[3][3], colB[3][3];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { colA[i][k] = RA[3*k + i]; colB[i][k] = RB[3*k + i]; }
    T best = (T)-1e30; int code = -1; T bestn[3] = {0, 0, 0};
    for (int i = 0; i < 6; i++) {
      T L[3];
      for (int k = 0; k < 3; k++) L[k] = i < 3 ? colA[i % 3][k] : colB[i % 3][k];
      T ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += sA[k]*t_abs(dot3(L, colA[k])); rb += sB[k]*t_abs(dot3(L, colB[k])); }
      const T proj = dot3(L, d), sep = t_abs(proj) - ra - rb;
      if (sep > margin) return 0;
      if (sep > best) { best = sep; code = i; for (int k = 0; k < 3; k++) bestn[k] = proj >= 0 ? L[k] : -L[k]; }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      T L[3];
      cross3(L, colA[i], colB[j]);
      const T ln = t_sqrt(dot3(L, L));
      if (ln < (T)1e-6) continue;
      for (int k = 0; k < 3; k++) L[k] /= ln;
      T ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += sA[k]*t_abs(dot3(L, colA[k])); rb += sB[k]*t_abs(dot3(L, colB[k])); }
      const T proj = dot3(L, d), sep = t_abs(proj) - ra - rb;
      if (sep > margin) return 0;
      if (sep > 0 ? sep > best : sep*(T)1.05 > best) {
        if (!(sep > 0) && !(best < 0)) continue;
        best = sep; code = 6 + 3*i + j; for (int k = 0; k < 3; k++) bestn[k] = proj >= 0 ? L[k] : -L[k];
      }
    }
    if (code >= 6) {
      const int i = (code - 6)/3, j = (code - 6) % 3;
      T pa[3] = {pA[0], pA[1], pA[2]}, pb[3] = {pB[0], pB[1], pB[2]};
      for (int k = 0; k < 3; k++) if (k != i) { const T sg = dot3(bestn, colA[k]) > 0 ? (T)1 : (T)-1; for (int a = 0; a < 3; a++) pa[a] += sg*sA[k]*colA[k][a]; }
      for (int k = 0; k < 3; k++) if (k != j) { const T sg = dot3(bestn, colB[k]) > 0 ? (T)-1 : (T)1; for (int a = 0; a < 3; a++) pb[a] += sg*sB[k]*colB[k][a]; }
      T ua[3], ub[3];
      for (int k = 0; k < 3; k++) { ua[k] = colA[i][k]; ub[k] = colB[j][k]; }
      const T w[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
      const T uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), den = 1 - uaub*uaub;
      T alpha = 0, beta = 0;
      if (den > (T)1e-12) { alpha = (q1 + uaub*q2)/den; beta = (uaub*q1 + q2)/den; }
      alpha = t_max(-sA[i], t_min(sA[i], alpha)); beta = t_max(-sB[j], t_min(sB[j], beta));
      hs->s0.dist = best;
      for (int k = 0; k < 3; k++) { hs->s0.pos[k] = (T)0.5*((pa[k] + alpha*ua[k]) + (pb[k] + beta*ub[k])); hs->s0.nrm[k] = bestn[k]; }
      return 1;
    }
    const bool refA = code < 3;
    T pR[3], sR[3], pI[3], sI[3], cR[3][3], cI[3][3], nref[3];
    for (int k = 0; k < 3; k++) {
      pR[k] = refA ? pA[k] : pB[k]; sR[k] = refA ? sA[k] : sB[k]; pI[k] = refA ? pB[k] : pA[k]; sI[k] = refA ? sB[k] : sA[k];
      nref[k] = refA ? bestn[k] : -bestn[k];
      for (int a = 0; a < 3; a++) { cR[k][a] = refA ? colA[k][a] : colB[k][a]; cI[k][a] = refA ? colB[k][a] : colA[k][a]; }
    }
    const int ax = refA ? code : code - 3;
    int inc = 0; T incdot = (T)1e30;
    for (int k = 0; k < 3; k++) { const T dk = dot3(nref, cI[k]); if (-t_abs(dk) < incdot) { incdot = -t_abs(dk); inc = k; } }
    const T incsign = dot3(nref, cI[inc]) > 0 ? (T)-1 : (T)1;
    const int i1 = (inc + 1) % 3, i2 = (inc + 2) % 3, r1 = (ax + 1) % 3, r2 = (ax + 2) % 3;
    // (a quad clipped by four half-planes has at most eight vertices: each clip of a convex polygon adds at most one.  The
    // arrays are indexed at run time, i.e. they live in scratch memory: 2 x 8 x 3 reals, half of what 16 slots took; the
    // guards below only matter if rounding ever made a clipped polygon non-convex)
    T poly[8][3], tmp[8][3];
    int np_ = 4;
    for (int v = 0; v < 4; v++) {
      const T a = (v == 0 || v == 3) ? (T)1 : (T)-1, b = v < 2 ? (T)1 : (T)-1;
      T pt[3];
      for (int k = 0; k < 3; k++) pt[k] = pI[k] + incsign*sI[inc]*cI[inc][k] + a*sI[i1]*cI[i1][k] + b*sI[i2]*cI[i2][k] - pR[k];
      poly[v][0] = dot3(pt, cR[r1]); poly[v][1] = dot3(pt, cR[r2]); poly[v][2] = dot3(pt, nref) - sR[ax];
    }
    for (int side = 0; side < 4 && np_ > 0; side++) {
      const int coord = side >> 1; const T sg = (side & 1) ? (T)-1 : (T)1, lim = coord ? sR[r2] : sR[r1];
      int nn = 0;
      for (int v = 0; v < np_; v++) {
        const T* P = poly[v]; const T* Q = poly[(v + 1) % np_];
        const T dp = lim - sg*P[coord], dq = lim - sg*Q[coord];
        if (dp >= 0 && nn < 8) { tmp[nn][0] = P[0]; tmp[nn][1] = P[1]; tmp[nn][2] = P[2]; nn++; }
        if ((dp >= 0) != (dq >= 0) && nn < 8) { const T f = dp/(dp - dq); for (int k = 0; k < 3; k++) tmp[nn][k] = P[k] + f*(Q[k] - P[k]); nn++; }
      }
      np_ = nn;
      for (int v = 0; v < np_; v++) for (int k = 0; k < 3; k++) poly[v][k] = tmp[v][k];
    }
    int nk = 0;
    for (int v = 0; v < np_; v++) if (poly[v][2] <= margin) { for (int k = 0; k < 3; k++) poly[nk][k] = poly[v][k]; nk++; }
    if (!nk) return 0;
    int pick[4] = {0, 0, 0, 0}, npick = 0;
    if (nk <= 4) { for (int v = 0; v < nk; v++) pick[npick++] = v; }
    else {
      T cx = 0, cy = 0; int deep = 0;
      for (int v = 0; v < nk; v++) { cx += poly[v][0]; cy += poly[v][1]; if (poly[v][2] < poly[deep][2]) deep = v; }
      cx /= nk; cy /= nk;
      const T PI = (T)3.14159265358979323846;
      const T a0 = t_atan2(poly[deep][1] - cy, poly[deep][0] - cx);
      int used = 1 << deep;
      pick[npick++] = deep;
      for (int q = 1; q < 4; q++) {
        const T target = a0 + q*(PI/2);
        int bv = -1; T bd = (T)1e30;
        for (int v = 0; v < nk; v++) if (!((used >> v) & 1)) {
          const T da = t_abs(t_fmod(t_atan2(poly[v][1] - cy, poly[v][0] - cx) - target + 5*PI, 2*PI) - PI);
          if (da < bd) { bd = da; bv = v; }
        }
        pick[npick++] = bv; used |= 1 << bv;
      }
    }
    for (int q = 0; q < npick; q++) {
      const T* P = poly[pick[q]];
      Hit x;
      x.dist = P[2];
      for (int k = 0; k < 3; k++) { x.pos[k] = pR[k] + P[0]*cR[r1][k] + P[1]*cR[r2][k] + (sR[ax] + (T)0.5*P[2])*nref[k]; x.nrm[k] = bestn[k]; }
      put_hit(hs, q, x);
    }
    return (1 << npick) - 1;
  }
  // narrow phase for one pair; returns the mask of valid slots of h[0..3]
  // (slot order = MuJoCo's contact order); tang = optional shared tangent
  // ---- sphere / capsule against a cylinder (the oracle's point_cylinder / sphere_cylinder_core / collide_capsule_cylinder) ----
  // closest point of the SOLID cylinder (centre p, unit axis a, radius R, half-height H) to q; returns the distance
  DMC_DEV static T point_cylinder(const T* q, const T* p, const T* a, T R, T H, T* closest) {
    const T v[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
    const T x = dot3(v, a), perp[3] = {v[0] - x*a[0], v[1] - x*a[1], v[2] - x*a[2]};
    const T d = t_sqrt(dot3(perp, perp)), xc = t_max(-H, t_min(H, x)), sc = d > R ? R/d : (T)1;
    for (int k = 0; k < 3; k++) closest[k] = p[k] + xc*a[k] + sc*perp[k];
    const T dif[3] = {q[0] - closest[0], q[1] - closest[1], q[2] - closest[2]};
    return t_sqrt(dot3(dif, dif));
  }
  DMC_DEV static int sphere_cylinder_core(Hit* hit, T margin, const T* ps, T rs, const T* p2, const T* m2, const T* s2) {
    const T a[3] = {m2[2], m2[5], m2[8]}, R = s2[0], H = s2[1];
    T closest[3], n[3], dist;
    const T g = point_cylinder(ps, p2, a, R, H, closest);
    if (g >= (T)DMC_MINVAL) {
      dist = g - rs;
      for (int k = 0; k < 3; k++) n[k] = (closest[k] - ps[k]) / g;
    } else {      // centre inside the solid: out through the nearest face
      const T v[3] = {ps[0] - p2[0], ps[1] - p2[1], ps[2] - p2[2]};
      const T x = dot3(v, a), perp[3] = {v[0] - x*a[0], v[1] - x*a[1], v[2] - x*a[2]}, d = t_sqrt(dot3(perp, perp));
      if (H - t_abs(x) < R - d) { dist = -(H - t_abs(x)) - rs; for (int k = 0; k < 3; k++) n[k] = x >= 0 ? -a[k] : a[k]; }
      else {
        dist = -(R - d) - rs;
        if (d < (T)DMC_MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else for (int k = 0; k < 3; k++) n[k] = -perp[k] / d;
      }
    }
    if (dist > margin) return 0;
    hit->dist = dist;
    for (int k = 0; k < 3; k++) { hit->pos[k] = ps[k] + n[k]*(rs + dist*(T)0.5); hit->nrm[k] = n[k]; }
    return 1;
  }
  // slope of the point-to-cylinder distance along the capsule axis at p1 + t u (nondecreasing in t)
  DMC_DEV static T segment_slope(T t, const T* p1, const T* u, const T* p2, const T* a, T R, T H) {
    const T q[3] = {p1[0] + t*u[0], p1[1] + t*u[1], p1[2] + t*u[2]};
    T closest[3];
    const T g = point_cylinder(q, p2, a, R, H, closest);
    if (g < (T)DMC_MINVAL) return 0;
    return ((q[0] - closest[0])*u[0] + (q[1] - closest[1])*u[1] + (q[2] - closest[2])*u[2]) / g;
  }
  DMC_DEV static T slope_crossing(T thr, T h, const T* p1, const T* u, const T* p2, const T* a, T R, T H) {
    if (segment_slope(-h, p1, u, p2, a, R, H) > thr) return -h;
    if (!(segment_slope(h, p1, u, p2, a, R, H) > thr)) return h;
    T lo = -h, hi = h;
    for (int it = 0; it < (sizeof(T) == 4 ? 28 : 60); it++) {
      const T t = (T)0.5*(lo + hi);
      if (segment_slope(t, p1, u, p2, a, R, H) > thr) hi = t; else lo = t;
    }
    return (T)0.5*(lo + hi);
  }
  // returns 0 / 1 contacts, or -1 when the capsule's axis reaches the cylinder (no unique closest pair: the caller warns)
  DMC_DEV static int capsule_cylinder(Hit* hit, T margin, const T* p1, const T* m1, const T* s1, const T* p2, const T* m2, const T* s2) {
    const T u[3] = {m1[2], m1[5], m1[8]}, a[3] = {m2[2], m2[5], m2[8]};
    const T tol = sizeof(T) == 4 ? (T)1e-4 : (T)1e-7;
    const T ta = slope_crossing(-tol, s1[1], p1, u, p2, a, s2[0], s2[1]);
    const T tb = slope_crossing(tol, s1[1], p1, u, p2, a, s2[0], s2[1]);
    const T t = (T)0.5*(ta + tb);
    const T q[3] = {p1[0] + t*u[0], p1[1] + t*u[1], p1[2] + t*u[2]};
    T closest[3];
    if (point_cylinder(q, p2, a, s2[0], s2[1], closest) < (sizeof(T) == 4 ? (T)1e-5 : (T)1e-9)*(s2[0] + s2[1])) return -1;
    return sphere_cylinder_core(hit, margin, q, s1[0], p2, m2, s2);
  }
  DMC_DEV int narrow_phase(int g1, int g2, T margin, Hits* h, T* tang, bool* has_tang, bool* guard) {
    int t1 = MI(geom_type)[g1], t2 = MI(geom_type)[g2];
    // cylinders: against a plane, a sphere or a capsule the narrow phase is restated; every other pair is tested as the
    // cylinder's enclosing capsule (same radius / half-length) and a hit only raises DMC_WARN_COLLISION (see collision())
    const bool plane_cyl = t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_CYLINDER;
    const bool cyl_pair = L.d.ncylx && t2 == DMC_GEOM_CYLINDER && (t1 == DMC_GEOM_SPHERE || t1 == DMC_GEOM_CAPSULE);
    const bool cyl_capsule = cyl_pair && t1 == DMC_GEOM_CAPSULE;
    *guard = (t1 == DMC_GEOM_CYLINDER || t2 == DMC_GEOM_CYLINDER) && !plane_cyl && !cyl_pair;
    if (t1 == DMC_GEOM_CYLINDER) t1 = DMC_GEOM_CAPSULE;
    if (t2 == DMC_GEOM_CYLINDER) t2 = DMC_GEOM_CAPSULE;
    const T *p1 = S(geom_xpos) + 3*g1, *p2 = S(geom_xpos) + 3*g2;
    const T *m1 = SG(geom_xmat) + 9*g1, *m2 = SG(geom_xmat) + 9*g2;
    // sizes and bounding radii as private copies: shared model tables, or the environment's own values
    T s1[3], s2[3], rb1 = MR(geom_rbound)[g1], rb2 = MR(geom_rbound)[g2];
    for (int k = 0; k < 3; k++) { s1[k] = MR(geom_size)[3*g1 + k]; s2[k] = MR(geom_size)[3*g2 + k]; }
    if (o.eg_n) {
      const int k1 = o.eg_slot[g1], k2 = o.eg_slot[g2];
      const T* eg = (const T*)o.eg_data + SI(imisc)[IM_ENV];
      if (k1 >= 0) { for (int k = 0; k < 3; k++) s1[k] = eg[(size_t)(16*k1 + 12 + k)*o.eg_B]; rb1 = eg[(size_t)(16*k1 + 15)*o.eg_B]; }
      if (k2 >= 0) { for (int k = 0; k < 3; k++) s2[k] = eg[(size_t)(16*k2 + 12 + k)*o.eg_B]; rb2 = eg[(size_t)(16*k2 + 15)*o.eg_B]; }
    }
    *has_tang = false;
    if (t1 == DMC_GEOM_PLANE) {
      T nrm[3] = {m1[2], m1[5], m1[8]};
      T dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      if (dot3(dif, nrm) > rb2 + margin) return 0;
      if (plane_cyl) {
        // mjc_PlaneCylinder: deepest rim point of the near cap, the matching point of the far cap, two
        // more points of the near disc at +-120 degrees (same order as the oracle)
        T axis[3] = {m2[2], m2[5], m2[8]};
        T prjaxis = dot3(nrm, axis);
        if (prjaxis > 0) { axis[0] = -axis[0]; axis[1] = -axis[1]; axis[2] = -axis[2]; prjaxis = -prjaxis; }
        const T dist0 = dot3(dif, nrm);
        T vec[3] = {axis[0]*prjaxis - nrm[0], axis[1]*prjaxis - nrm[1], axis[2]*prjaxis - nrm[2]};
        const T len2 = dot3(vec, vec);
        if (len2 >= (T)DMC_MINVAL*(T)DMC_MINVAL) { const T sc = s2[0]/t_sqrt(len2); vec[0] *= sc; vec[1] *= sc; vec[2] *= sc; }
        else { vec[0] = m2[0]*s2[0]; vec[1] = m2[3]*s2[0]; vec[2] = m2[6]*s2[0]; }
        const T prjvec = dot3(vec, nrm);
        axis[0] *= s2[1]; axis[1] *= s2[1]; axis[2] *= s2[1]; prjaxis *= s2[1];
        if (dist0 + prjaxis + prjvec > margin) return 0;
        int cnt = 0;
        Hit x;
        for (int k = 0; k < 3; k++) x.nrm[k] = nrm[k];
        x.dist = dist0 + prjaxis + prjvec;
        for (int k = 0; k < 3; k++) x.pos[k] = p2[k] + vec[k] + axis[k] - nrm[k]*x.dist*(T)0.5;
        put_hit(h, cnt, x); cnt++;
        if (dist0 - prjaxis + prjvec <= margin) {
          x.dist = dist0 - prjaxis + prjvec;
          for (int k = 0; k < 3; k++) x.pos[k] = p2[k] + vec[k] - axis[k] - nrm[k]*x.dist*(T)0.5;
          put_hit(h, cnt, x); cnt++;
        }
        const T prjvec1 = -prjvec*(T)0.5;
        if (dist0 + prjaxis + prjvec1 <= margin) {
          T vec1[3];
          cross3(vec1, vec, axis);
          normalize3(vec1);
          const T sc = s2[0]*t_sqrt((T)3)/2;
          x.dist = dist0 + prjaxis + prjvec1;
          for (int k = 0; k < 3; k++) x.pos[k] = p2[k] + sc*vec1[k] + axis[k] - vec[k]*(T)0.5 - nrm[k]*x.dist*(T)0.5;
          put_hit(h, cnt, x); cnt++;
          for (int k = 0; k < 3; k++) x.pos[k] = p2[k] - sc*vec1[k] + axis[k] - vec[k]*(T)0.5 - nrm[k]*x.dist*(T)0.5;
          put_hit(h, cnt, x); cnt++;
        }
        return (1 << cnt) - 1;
      }
      if (t2 == DMC_GEOM_SPHERE) return plane_sphere(&h->s0, margin, p1, nrm, p2, s2[0]);
      if (t2 == DMC_GEOM_CAPSULE) {
        T axis[3] = {m2[2], m2[5], m2[8]};
        T seg[3] = {axis[0]*s2[1], axis[1]*s2[1], axis[2]*s2[1]}, pos[3];
        T dp = dot3(nrm, axis);
        for (int k = 0; k < 3; k++) tang[k] = axis[k] - nrm[k]*dp;
        normalize3(tang);
        *has_tang = true;
        int mask = 0;
        for (int k = 0; k < 3; k++) pos[k] = p2[k] + seg[k];
        mask |= plane_sphere(&h->s0, margin, p1, nrm, pos, s2[0]);
        for (int k = 0; k < 3; k++) pos[k] = p2[k] - seg[k];
        mask |= plane_sphere(&h->s1, margin, p1, nrm, pos, s2[0]) << 1;
        return mask;
      }
      if (t2 == DMC_GEOM_BOX) {
        T dist = dot3(dif, nrm);
        int cnt = 0;
        for (int i = 0; i < 8; i++) {
          T vec[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])}, corner[3];
          mul_mat_vec3(corner, m2, vec);
          T ldist = dot3(nrm, corner);
          if (dist + ldist > margin || ldist > 0 || cnt >= 4) continue;
          Hit x;
          x.dist = dist + ldist;
          for (int k = 0; k < 3; k++) { x.nrm[k] = nrm[k]; x.pos[k] = corner[k] + p2[k] - nrm[k]*x.dist*(T)0.5; }
          put_hit(h, cnt, x);
          cnt++;
        }
        return (1 << cnt) - 1;
      }
      if (L.d.nell && t2 == DMC_GEOM_ELLIPSOID) return plane_ellipsoid(&h->s0, margin, p1, nrm, p2, m2, s2);
      return 0;
    }
    {
      T dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      T bound = rb1 + rb2 + margin;
      if (dot3(dif, dif) > bound*bound) return 0;
    }
    if (L.d.ncylx && cyl_pair) {
      if (!cyl_capsule) return sphere_cylinder_core(&h->s0, margin, p1, s1[0], p2, m2, s2);
      const int r = capsule_cylinder(&h->s0, margin, p1, m1, s1, p2, m2, s2);
      if (r < 0) { *guard = true; return 1; }      // the axis reaches the cylinder: DMC_WARN_COLLISION, no contact
      return r;
    }
    if (L.d.nell && t2 == DMC_GEOM_ELLIPSOID) return ellipsoid_pair(&h->s0, margin, t1, p1, m1, s1, p2, m2, s2);
    if (L.d.nbox && t2 == DMC_GEOM_BOX) {
      if (t1 == DMC_GEOM_SPHERE) return sphere_box_core(&h->s0, margin, p1, s1[0], p2, m2, s2);
      if (t1 == DMC_GEOM_CAPSULE) return capsule_box(h, margin, p1, m1, s1, p2, m2, s2);
      if (t1 == DMC_GEOM_BOX) return box_box(h, margin, p1, m1, s1, p2, m2, s2);
    }
    if (t1 == DMC_GEOM_SPHERE && t2 == DMC_GEOM_SPHERE) return sphere_sphere(&h->s0, margin, p1, s1[0], p2, s2[0]);
    if (t1 == DMC_GEOM_SPHERE && t2 == DMC_GEOM_CAPSULE) {
      T axis[3] = {m2[2], m2[5], m2[8]}, vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
      T x = dot3(axis, vec);
      x = t_max(-s2[1], t_min(s2[1], x));
      T q[3] = {p2[0] + axis[0]*x, p2[1] + axis[1]*x, p2[2] + axis[2]*x};
      return sphere_sphere(&h->s0, margin, p1, s1[0], q, s2[0]);
    }
    if (t1 == DMC_GEOM_CAPSULE && t2 == DMC_GEOM_CAPSULE) {
      T a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
      T dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
      T ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
      T u = -dot3(a1, dif), v = dot3(a2, dif), det = ma*mc - mb*mb;
      T v1[3], v2[3];
      if (t_abs(det) >= (T)DMC_MINVAL) {
        T x1 = (mc*u - mb*v) / det, x2 = (ma*v - mb*u) / det;
        if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb*s1[1]) / mc; }
        else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v + mb*s1[1]) / mc; }
        if (x2 > s2[1]) { x2 = s2[1]; x1 = (u - mb*s2[1]) / ma; x1 = t_max(-s1[1], t_min(s1[1], x1)); }
        else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = (u + mb*s2[1]) / ma; x1 = t_max(-s1[1], t_min(s1[1], x1)); }
        for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k]*x1; v2[k] = p2[k] + a2[k]*x2; }
        return sphere_sphere(&h->s0, margin, v1, s1[0], v2, s2[0]);
      }
      int n = 0;
      for (int sg = 1; sg >= -1 && n < 2; sg -= 2) {
        T x1 = sg * s1[1], x2 = (v - mb*x1) / mc;
        if (x2 >= -s2[1] && x2 <= s2[1]) {
          for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k]*x1; v2[k] = p2[k] + a2[k]*x2; }
          Hit x;
          if (sphere_sphere(&x, margin, v1, s1[0], v2, s2[0])) { put_hit(h, n, x); n++; }
        }
      }
      for (int sg = 1; sg >= -1 && n < 2; sg -= 2) {
        T x2 = sg * s2[1], x1 = (u - mb*x2) / ma;
        if (x1 > -s1[1] && x1 < s1[1]) {
          for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k]*x1; v2[k] = p2[k] + a2[k]*x2; }
          Hit x;
          if (sphere_sphere(&x, margin, v1, s1[0], v2, s2[0])) { put_hit(h, n, x); n++; }
        }
      }
      return (1 << n) - 1;
    }
    return 0;
  }
  // contact-parameter tuple of a candidate pair; a model whose pairs all share one tuple (cheetah,
  // walker, ... in their specialised kernels) needs no lookup at all
  DMC_DEV int prm_of_info(int info) const { return L.d.nprm == 1 ? 0 : (int)((unsigned)info >> 8); }
  // per-contact copies of the candidate pair's entries (the pair tables themselves stay in global memory)
  DMC_DEV int con_g1(int c) const { return SI(con_geom)[c] & 0xffff; }
  DMC_DEV int con_g2(int c) const { return (int)((unsigned)SI(con_geom)[c] >> 16); }
  DMC_DEV int con_b1(int c) const { return MI(geom_bodyid)[con_g1(c)]; }
  DMC_DEV int con_b2(int c) const { return MI(geom_bodyid)[con_g2(c)]; }
  DMC_DEV int con_dim(int c) const { return SI(con_info)[c] & 0xff; }
  DMC_DEV int con_prm(int c) const { return prm_of_info(SI(con_info)[c]); }
  DMC_DEV void collision() {
    int base = 0;
    const bool enabled = !(o.disableflags & (DMC_DSBL_CONTACT | DMC_DSBL_CONSTRAINT));
    const int npair = enabled ? L.d.npair : 0;
    int overflow = 0, unresolved = 0;
    // the candidate pair list lives in global memory: each trip's entries are fetched one trip ahead
    int nxt_geom = lane < npair ? GC(pair_geom)[lane] : 0, nxt_info = lane < npair ? GC(pair_info)[lane] : 0;
    for (int p0 = 0; p0 < npair; p0 += LPE) {
      const int p = p0 + lane;
      Hits h = {}; T tang[3] = {0, 0, 0}; bool has_tang = false;
      int mask = 0;
      const int pgeom = nxt_geom, pinfo = nxt_info;
      if (p + LPE < npair) { nxt_geom = GC(pair_geom)[p + LPE]; nxt_info = GC(pair_info)[p + LPE]; }
      if (p < npair) {
        bool guard;
        mask = narrow_phase(pgeom & 0xffff, (int)((unsigned)pgeom >> 16), MR(prm_margin)[prm_of_info(pinfo)], &h, tang, &has_tang, &guard);
        if (guard) { if (mask) unresolved = 1; mask = 0; }
      }
      const int n = __builtin_popcount(mask);
      int total;
      int off = base + group_scan<LPE>(n, lane, &total);
      for (int i = 0; i < 4; i++) if ((mask >> i) & 1) {
        const int c = off + __builtin_popcount(mask & ((1 << i) - 1));
        if (c >= L.d.nconmax) { overflow = 1; continue; }
        const Hit& hi = i == 0 ? h.s0 : (i == 1 ? h.s1 : (i == 2 ? h.s2 : h.s3));   // i is a constant after unrolling
        T f[9];
        for (int k = 0; k < 3; k++) { f[k] = hi.nrm[k]; f[3 + k] = has_tang ? tang[k] : (T)0; }
        if (has_tang) cross3(f + 6, f, f + 3); else make_frame(f);
        S(con_dist)[c] = hi.dist;
        for (int k = 0; k < 3; k++) S(con_pos)[3*c + k] = hi.pos[k];
        for (int k = 0; k < 9; k++) conF()[9*c + k] = f[k];
        SI(con_geom)[c] = pgeom; SI(con_info)[c] = pinfo;
        SI(con_efc)[c] = -1;
      }
      base += total;
    }
    overflow = group_max<LPE>(overflow);
    if (L.d.ncyl) unresolved = group_max<LPE>(unresolved);
    if (base > L.d.nconmax) base = L.d.nconmax;
    if (lane == 0) {
      SI(imisc)[IM_NCON] = base;
      if (overflow) SI(imisc)[IM_WARN + DMC_WARN_CONTACTFULL]++;
      if (unresolved) SI(imisc)[IM_WARN + DMC_WARN_COLLISION]++;
    }
    DMC_WSYNC();
  }

  // ---- constraint assembly (mj_makeConstraint + impedance) ----------------------
  DMC_DEV static T get_impedance(const T* si_, T pos, T margin) {
    T s0 = t_max((T)DMC_MINIMP, t_min((T)DMC_MAXIMP, si_[0]));
    T s1 = t_max((T)DMC_MINIMP, t_min((T)DMC_MAXIMP, si_[1]));
    T s2 = t_max((T)0, si_[2]);
    T s3 = t_max((T)DMC_MINIMP, t_min((T)DMC_MAXIMP, si_[3]));
    T s4 = t_max((T)1, si_[4]);
    if (s0 == s1 || s2 <= (T)DMC_MINVAL) return (T)0.5*(s0 + s1);
    T x = (pos - margin) / s2;
    if (x < 0) x = -x;
    if (x >= 1) return s1;
    if (x == 0) return s0;
    T y;
    if (s4 == 1) y = x;
    // power 2 is MuJoCo's default solimp: squares instead of four pow() expansions (both branches run when lanes diverge)
    else if (s4 == 2) y = x <= s3 ? (1 / s3) * (x*x) : 1 - (1 / (1 - s3)) * ((1 - x)*(1 - x));
    else if (x <= s3) y = (1 / t_pow(s3, s4 - 1)) * t_pow(x, s4);
    else y = 1 - (1 / t_pow(1 - s3, s4 - 1)) * t_pow(1 - x, s4);
    return s0 + y*(s1 - s0);
  }
  DMC_DEV bool dof_in_chain(int lastdof, int dofid) const {
    if (lastdof < 0) return false;
    const unsigned m = (unsigned)(dofid < 32 ? MI(dof_anc_lo)[lastdof] : MI(dof_anc_hi)[lastdof]);
    return (m >> (dofid & 31)) & 1u;
  }
  // models with elliptic cones or dof friction loss take the general (per-row-type) solver paths
  DMC_DEV bool general_rows() const { return L.d.elliptic || L.d.nfric || L.d.neq; }
  // ---- tendons (mj_tendon for fixed and site-to-site spatial tendons) -----------------------
  DMC_DEV void site_world_pos(int sid, T* p) {
    const int b = MI(site_bodyid)[sid]; T v[3];
    mul_mat_vec3(v, S(xmat) + 9*b, MRS(site_pos) + 3*sid);
    for (int k = 0; k < 3; k++) p[k] = S(xpos)[3*b + k] + v[k];
  }
  DMC_DEV T tendon_length(int t) {
    const int w0 = MI(tendon_adr)[t], wn = MI(tendon_num)[t];
    T len = 0;
    if (MI(wrap_site)[w0] < 0) { for (int w = w0; w < w0 + wn; w++) len += MR(wrap_prm)[w] * S(qpos)[MI(wrap_qpos)[w]]; return len; }
    for (int w = w0; w + 1 < w0 + wn; w++) {
      T p0[3], p1[3];
      site_world_pos(MI(wrap_site)[w], p0); site_world_pos(MI(wrap_site)[w + 1], p1);
      const T dif[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
      len += t_sqrt(dot3(dif, dif));
    }
    return len;
  }
  // translational Jacobian column of a world point rigidly attached to `body`
  DMC_DEV void point_jac(int body, const T* p, int dd, T* jp) {
    jp[0] = jp[1] = jp[2] = 0;
    if (!dof_in_chain(MI(body_lastdof)[body], dd)) return;
    const T* cd = S(cdof) + 6*dd; const T* rc = S(subtree_com) + 3*MI(body_rootid)[body];
    T off[3] = {p[0] - rc[0], p[1] - rc[1], p[2] - rc[2]}, tmp[3];
    cross3(tmp, cd, off);
    for (int k = 0; k < 3; k++) jp[k] = cd[3 + k] + tmp[k];
  }
  DMC_DEV T tendon_jac(int t, int dd) {
    const int w0 = MI(tendon_adr)[t], wn = MI(tendon_num)[t];
    T j = 0;
    if (MI(wrap_site)[w0] < 0) { for (int w = w0; w < w0 + wn; w++) if (MI(wrap_dof)[w] == dd) j += MR(wrap_prm)[w]; return j; }
    for (int w = w0; w + 1 < w0 + wn; w++) {
      const int s0 = MI(wrap_site)[w], s1 = MI(wrap_site)[w + 1];
      T p0[3], p1[3], j0[3], j1[3];
      site_world_pos(s0, p0); site_world_pos(s1, p1);
      T dif[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
      const T n = t_sqrt(dot3(dif, dif));
      if (n < (T)DMC_MINVAL) continue;
      for (int k = 0; k < 3; k++) dif[k] /= n;
      point_jac(MI(site_bodyid)[s0], p0, dd, j0); point_jac(MI(site_bodyid)[s1], p1, dd, j1);
      j += dif[0]*(j1[0] - j0[0]) + dif[1]*(j1[1] - j0[1]) + dif[2]*(j1[2] - j0[2]);
    }
    return j;
  }
  DMC_DEV int contact_rows(int dim) const { return dim == 1 ? 1 : (L.d.elliptic ? dim : 2*(dim - 1)); }
  // ---- constraint Jacobian access (three storage classes, see step_layout.h) -------------------
  struct RowMap { int s0, tl0, c0; };   // first simple / tendon-limit / contact row (group-uniform)
  DMC_DEV RowMap row_map() const { RowMap rm = {SI(imisc)[IM_ROW_S0], SI(imisc)[IM_ROW_TL0], SI(imisc)[IM_ROW_C0]}; return rm; }
  // the one nonzero of a dof-friction / joint-limit row
  DMC_DEV int simple_dof(int tid) const { return EFC_TYPE(tid) == EFC_FRICTION ? EFC_ID(tid) : (EFC_ID(tid) >> 1); }
  DMC_DEV T simple_sign(int tid) const { return (EFC_TYPE(tid) == EFC_LIMIT && (EFC_ID(tid) & 1)) ? (T)-1 : (T)1; }
  // the compressed contact rows: LDS, or the global scratch of a large model.  Model-specialised kernels know which
  // at compile time and get an address-space-qualified pointer (ds_* or global_* instead of flat_* accesses).
  DMC_DEV auto Jc() const {
    if constexpr (LS::kJGlobal >= 1) return (DMC_GLB T*)(gscr() + L.gs_Jc);
    else if constexpr (LS::kJGlobal == 0) return (DMC_LDS T*)S(efc_Jc);
    else return L.d.jglobal ? gscr() + L.gs_Jc : S(efc_Jc);
  }
  DMC_DEV const T* dense_row(int r, const RowMap& rm) const { return S(efc_Jd) + (r < rm.s0 ? r : rm.s0 + (r - rm.tl0))*L.d.nv; }
  DMC_DEV const unsigned char* con_dof_list(int c) const { return (const unsigned char*)(SI(con_dofs) + c*L.d.kwords); }
  DMC_DEV int con_ndof(int c) const { return __builtin_popcount(con_mask_lo(c)) + (L.d.nv > 32 ? __builtin_popcount(con_mask_hi(c)) : 0); }
  DMC_DEV unsigned con_mask_lo(int c) const { return (unsigned)SI(con_mlo)[c]; }
  DMC_DEV unsigned con_mask_hi(int c) const { return L.d.nv > 32 ? (unsigned)SI(con_mhi)[c] : 0u; }
  // slot of dof dd in the compressed rows of a contact with mask (lo, hi), or -1
  DMC_DEV int mask_slot(unsigned lo, unsigned hi, int dd) const {
    if (L.d.nv <= 32 || dd < 32) {
      if (!((lo >> dd) & 1u)) return -1;
      return __builtin_popcount(lo & ((1u << dd) - 1u));
    }
    const int d2 = dd - 32;
    if (!((hi >> d2) & 1u)) return -1;
    return __builtin_popcount(lo) + __builtin_popcount(hi & ((1u << d2) - 1u));
  }
  // J[r, :] . x for any row class (x: nv reals)
  DMC_DEV T row_dot(int r, const T* x, const RowMap& rm) const {
    if (L.d.jfull) return dot_n(S(efc_Jd) + r*L.d.nv, x, L.d.nv);
    if (r >= rm.c0) {
      // entry k of the row belongs to dof con_dofs[c][k]; the loop runs to the compile-time bound kmax in the
      // model-specialised kernels, so all its loads are in flight together
      const int c = EFC_ID(SI(efc_tid)[r]);
      const auto jr = Jc() + (r - rm.c0)*L.d.kmax;
      const unsigned char* dofs = con_dof_list(c);
      const int kc = con_ndof(c);
      T acc = 0;
      // (the slots past the contact's dof count are read like the others -- they exist, their dof byte addresses some word of
      // the scratch -- and their product is dropped: no exec-mask sequence per slot)
#ifdef DMC_HOST_EMU
      for (int k = 0; k < L.d.kmax; k++) { const T j_ = jr[k], x_ = x[k < kc ? dofs[k] : 0], t_ = acc + j_ * x_; acc = k < kc ? t_ : acc; }      // (the host build stays inside x)
#else
      for (int k = 0; k < L.d.kmax; k++) { const T j_ = jr[k], x_ = x[dofs[k]], t_ = acc + j_ * x_; acc = k < kc ? t_ : acc; }
#endif
      return acc;
    }
    if (r >= rm.s0 && r < rm.tl0) { const int tid = SI(efc_tid)[r]; return simple_sign(tid) * x[simple_dof(tid)]; }
    return dot_n(dense_row(r, rm), x, L.d.nv);
  }
  // J[r, dd] for any row class
  DMC_DEV T row_entry(int r, int dd, const RowMap& rm) const {
    if (L.d.jfull) return S(efc_Jd)[r*L.d.nv + dd];
    if (r >= rm.c0) {
      const int c = EFC_ID(SI(efc_tid)[r]);
      const int k = mask_slot(con_mask_lo(c), con_mask_hi(c), dd);
      return k < 0 ? (T)0 : Jc()[(r - rm.c0)*L.d.kmax + k];
    }
    if (r >= rm.s0 && r < rm.tl0) { const int tid = SI(efc_tid)[r]; return simple_dof(tid) == dd ? simple_sign(tid) : (T)0; }
    return dense_row(r, rm)[dd];
  }
  DMC_DEV void make_constraint() {
    const int nv = L.d.nv, njmax = L.d.njmax;
    int nefc = 0, overflow = 0, ndense = 0;
    const bool enabled = !(o.disableflags & DMC_DSBL_CONSTRAINT);
    // equality constraints first (MuJoCo order: equality, friction, limit, contact), two-sided, always
    // active; residuals and Jacobians as in the oracle's make_constraint (tendon, joint, connect, weld)
    if (L.d.neq && enabled && !(o.disableflags & DMC_DSBL_EQUALITY)) for (int k = 0; k < L.d.neq; k++) {
      const int et = MI(eq_type)[k], o1 = MI(eq_obj1)[k], o2 = MI(eq_obj2)[k];
      const int nrow = et == DMC_EQ_CONNECT ? 3 : (et == DMC_EQ_WELD ? 6 : 1);
      if (nefc + nrow > njmax || ndense + nrow > L.d.njdense) { overflow = 1; continue; }
      const int r0 = nefc;
      nefc += nrow; ndense += nrow;
      const T* data = MR(eq_data) + 13*k;
      T* Jd = S(efc_Jd);
      if (lane == 0) for (int a = 0; a < nrow; a++) { S(efc_D)[r0 + a] = 0; SI(efc_tid)[r0 + a] = EFC_TID(EFC_EQUALITY, k); }
      if (et == DMC_EQ_TENDON) {
        FOR_LANES(dd, nv) Jd[r0*nv + dd] = tendon_jac(o1, dd);
        if (lane == 0) S(efc_aref)[r0] = tendon_length(o1) - data[11] - data[0];
      } else if (et == DMC_EQ_JOINT) {
        const int q1 = MI(jnt_qposadr)[o1], d1 = MI(jnt_dofadr)[o1];
        T pos = S(qpos)[q1] - data[11], deriv = 0;
        int d2 = -1;
        if (o2 >= 0) {
          d2 = MI(jnt_dofadr)[o2];
          const T dif = S(qpos)[MI(jnt_qposadr)[o2]] - data[12];
          T pw = 1, poly = 0;
          for (int p = 0; p < 5; p++) { poly += data[p]*pw; if (p < 4) deriv += (p + 1)*data[p + 1]*pw; pw *= dif; }
          pos -= poly;
        } else pos -= data[0];
        FOR_LANES(dd, nv) { T j = dd == d2 ? -deriv : (T)0; if (dd == d1) j += 1; Jd[r0*nv + dd] = j; }
        if (lane == 0) S(efc_aref)[r0] = pos;
      } else {
        // connect: data[0:3] on body 1, data[3:6] on body 2; weld: data[3:6] on body 1, data[0:3] on body 2
        const T *l1 = et == DMC_EQ_CONNECT ? data : data + 3, *l2 = et == DMC_EQ_CONNECT ? data + 3 : data;
        T p1[3], p2[3], tmp[3];
        mul_mat_vec3(tmp, S(xmat) + 9*o1, l1); for (int a = 0; a < 3; a++) p1[a] = S(xpos)[3*o1 + a] + tmp[a];
        mul_mat_vec3(tmp, S(xmat) + 9*o2, l2); for (int a = 0; a < 3; a++) p2[a] = S(xpos)[3*o2 + a] + tmp[a];
        T quat[4] = {1, 0, 0, 0}, q2inv[4] = {1, 0, 0, 0}, err[4] = {1, 0, 0, 0};
        if (et == DMC_EQ_WELD) {
          T rel[4] = {data[6], data[7], data[8], data[9]}, q1[4], q2[4];
          for (int a = 0; a < 4; a++) { q1[a] = SG(xquat)[4*o1 + a]; q2[a] = SG(xquat)[4*o2 + a]; }
          mul_quat(quat, q1, rel);
          q2inv[0] = q2[0]; q2inv[1] = -q2[1]; q2inv[2] = -q2[2]; q2inv[3] = -q2[3];
          mul_quat(err, q2inv, quat);
        }
        if (lane == 0) {
          for (int a = 0; a < 3; a++) S(efc_aref)[r0 + a] = p1[a] - p2[a];
          if (et == DMC_EQ_WELD) for (int a = 0; a < 3; a++) S(efc_aref)[r0 + 3 + a] = data[10]*err[1 + a];
        }
        FOR_LANES(dd, nv) {
          T jp1[3], jp2[3];
          point_jac(o1, p1, dd, jp1); point_jac(o2, p2, dd, jp2);
          for (int a = 0; a < 3; a++) Jd[(r0 + a)*nv + dd] = jp1[a] - jp2[a];
          if (et == DMC_EQ_WELD) {
            const bool in1 = dof_in_chain(MI(body_lastdof)[o1], dd), in2 = dof_in_chain(MI(body_lastdof)[o2], dd);
            const T* cd = S(cdof) + 6*dd;
            T w[4] = {0, 0, 0, 0};
            for (int a = 0; a < 3; a++) w[1 + a] = (in1 ? cd[a] : (T)0) - (in2 ? cd[a] : (T)0);
            T t1[4], t2[4];
            mul_quat(t1, q2inv, w);
            mul_quat(t2, t1, quat);
            for (int a = 0; a < 3; a++) Jd[(r0 + 3 + a)*nv + dd] = data[10]*(T)0.5*t2[1 + a];
          }
        }
      }
    }
    const int row_s0 = nefc;
    // dof friction loss: one row per dof with frictionloss > 0 (simple rows: no Jacobian storage)
    if (L.d.nfric && enabled && !(o.disableflags & DMC_DSBL_FRICTIONLOSS)) {
      const int base = nefc;
      for (int k = lane; k < L.d.nfric; k += LPE) {
        const int r = base + k;
        if (r >= njmax) { overflow = 1; continue; }
        S(efc_aref)[r] = 0; S(efc_D)[r] = 0; SI(efc_tid)[r] = EFC_TID(EFC_FRICTION, MI(fric_dof)[k]);
        if (L.d.jfull) for (int k2 = 0; k2 < nv; k2++) S(efc_Jd)[r*nv + k2] = k2 == MI(fric_dof)[k] ? (T)1 : (T)0;
      }
      nefc = base + L.d.nfric < njmax ? base + L.d.nfric : njmax;
    }
    // joint limits (simple rows)
    if (enabled && !(o.disableflags & DMC_DSBL_LIMIT)) for (int j0 = 0; j0 < L.d.njnt; j0 += LPE) {
      const int j = j0 + lane;
      // lower side first, then upper (locals indexed statically: no scratch memory)
      bool act_lo = false, act_hi = false; T d_lo = 0, d_hi = 0, margin = 0;
      if (j < L.d.njnt && MI(jnt_limited)[j]) {
        const int t = MI(jnt_type)[j];
        if (t == DMC_JNT_SLIDE || t == DMC_JNT_HINGE) {
          const T value = S(qpos)[MI(jnt_qposadr)[j]];
          margin = MRC(jnt_margin)[j];
          d_lo = -(MRC(jnt_range)[2*j] - value); d_hi = MRC(jnt_range)[2*j + 1] - value;
          act_lo = d_lo < margin; act_hi = d_hi < margin;
        }
      }
      const int cnt = (act_lo ? 1 : 0) + (act_hi ? 1 : 0);
      int total;
      const int off = nefc + group_scan<LPE>(cnt, lane, &total);
      for (int i = 0; i < 2; i++) if (i < cnt) {
        const int r = off + i;
        if (r >= njmax) { overflow = 1; continue; }
        const bool lo = i == 0 && act_lo;
        S(efc_aref)[r] = lo ? d_lo : d_hi; S(efc_D)[r] = margin; SI(efc_tid)[r] = EFC_TID(EFC_LIMIT, (MI(jnt_dofadr)[j] << 1) | (lo ? 0 : 1));
        if (L.d.jfull) for (int k2 = 0; k2 < nv; k2++) S(efc_Jd)[r*nv + k2] = k2 == MI(jnt_dofadr)[j] ? (lo ? (T)1 : (T)-1) : (T)0;
      }
      nefc += total;
    }
    if (nefc > njmax) nefc = njmax;
    if (L.d.jfull) ndense = nefc;      // every row so far is a dense row: row r lives at efc_Jd[r * nv]
    const int row_tl0 = nefc;
    // tendon length limits (after the joint limits, as in MuJoCo); every lane evaluates the few
    // limited tendons' lengths itself, so the row count stays group-uniform without a fence
    if (L.d.nlimten && enabled && !(o.disableflags & DMC_DSBL_LIMIT)) for (int kt = 0; kt < L.d.nlimten; kt++) {
      const int t = MI(limten)[kt];
      const T value = tendon_length(t), margin = MR(tendon_margin)[t];
      for (int sd = -1; sd <= 1; sd += 2) {
        const T dist = sd * (MR(tendon_range)[2*t + (sd + 1)/2] - value);
        if (!(dist < margin)) continue;
        if (nefc >= njmax || ndense >= L.d.njdense) { overflow = 1; continue; }
        const int r = nefc++, jr = ndense++;
        FOR_LANES(dd, nv) S(efc_Jd)[jr*nv + dd] = -(T)sd * tendon_jac(t, dd);
        if (lane == 0) { S(efc_aref)[r] = dist; S(efc_D)[r] = margin; SI(efc_tid)[r] = EFC_TID(EFC_TENDON_LIMIT, t); }
      }
    }
    // ball-joint limits (mj_instantiateLimit, mjJNT_BALL): the rotation angle against max(range), one dense row with
    // J = -axis on the joint's three dofs.  Emitted here, after the tendon limits, because they need Jacobian storage
    // (MuJoCo and the oracle emit them in joint order: same rows, same minimiser); id = ntendon + joint
    if (L.d.nlimball && enabled && !(o.disableflags & DMC_DSBL_LIMIT)) for (int kb = 0; kb < L.d.nlimball; kb++) {
      const int j = MI(limball)[kb], qa = MI(jnt_qposadr)[j], da = MI(jnt_dofadr)[j];
      const T q0 = S(qpos)[qa], q1 = S(qpos)[qa + 1], q2 = S(qpos)[qa + 2], q3 = S(qpos)[qa + 3];
      const T qn = t_sqrt(q0*q0 + q1*q1 + q2*q2 + q3*q3);
      T ax[3] = {q1/qn, q2/qn, q3/qn};
      const T sn = t_sqrt(ax[0]*ax[0] + ax[1]*ax[1] + ax[2]*ax[2]);
      T angle = 2*t_atan2(sn, q0/qn);
      if (angle > (T)3.14159265358979323846) angle -= (T)6.283185307179586476925286766559;
      if (sn < (T)DMC_MINVAL) { ax[0] = ax[1] = ax[2] = 0; angle = 0; } else for (int k = 0; k < 3; k++) ax[k] /= sn;
      if (angle < 0) { angle = -angle; for (int k = 0; k < 3; k++) ax[k] = -ax[k]; }
      const T margin = MRC(jnt_margin)[j];
      const T dist = t_max(MRC(jnt_range)[2*j], MRC(jnt_range)[2*j + 1]) - angle;
      if (!(dist < margin)) continue;
      if (nefc >= njmax || ndense >= L.d.njdense) { overflow = 1; continue; }
      const int r = nefc++, jr = ndense++;
      FOR_LANES(dd, nv) S(efc_Jd)[jr*nv + dd] = dd == da ? -ax[0] : dd == da + 1 ? -ax[1] : dd == da + 2 ? -ax[2] : (T)0;      // (no dynamic index into the axis: no scratch)
      if (lane == 0) { S(efc_aref)[r] = dist; S(efc_D)[r] = margin; SI(efc_tid)[r] = EFC_TID(EFC_TENDON_LIMIT, L.d.ntendon + j); }
    }
    const int nefc_lim = nefc;
    // contact rows: headers, and the dof mask of each contact's Jacobian rows (the dofs on exactly one of
    // the two bodies' chains: on a shared ancestor dof the two bodies move together and the entry is 0)
    const int ncon = (enabled && !(o.disableflags & DMC_DSBL_CONTACT)) ? SI(imisc)[IM_NCON] : 0;
    for (int c0 = 0; c0 < ncon; c0 += LPE) {
      const int c = c0 + lane;
      int nrow = 0, dim = 0;
      T incl = 0;
      if (c < ncon) {
        dim = con_dim(c);
        incl = MR(prm_margin)[con_prm(c)] - MR(prm_gap)[con_prm(c)];
        nrow = contact_rows(dim);
        if (S(con_dist)[c] >= incl) nrow = 0;   // in the gap: excluded
      }
      int total;
      const int off = nefc + group_scan<LPE>(nrow, lane, &total);
      if (c < ncon) {
        if (nrow == 0) SI(con_efc)[c] = -1;
        else if (off + nrow > njmax || off - nefc_lim + nrow > L.d.njcon) { SI(con_efc)[c] = -1; overflow = 1; }
        else {
          SI(con_efc)[c] = off;
          const int l1 = MI(body_lastdof)[con_b1(c)], l2 = MI(body_lastdof)[con_b2(c)];
          if (!L.d.jfull) SI(con_mlo)[c] = (l1 >= 0 ? MI(dof_anc_lo)[l1] : 0) ^ (l2 >= 0 ? MI(dof_anc_lo)[l2] : 0);
          if (nv > 32) SI(con_mhi)[c] = (l1 >= 0 ? MI(dof_anc_hi)[l1] : 0) ^ (l2 >= 0 ? MI(dof_anc_hi)[l2] : 0);
          // pyramidal edges all carry (dist, margin); elliptic friction rows carry (0, 0)
          const bool ell = L.d.elliptic && dim > 1;
          for (int r = off; r < off + nrow; r++) {
            const bool nrm = !ell || r == off;
            S(efc_aref)[r] = nrm ? S(con_dist)[c] : (T)0; S(efc_D)[r] = nrm ? incl : (T)0;
            SI(efc_tid)[r] = EFC_TID(dim == 1 ? EFC_FRICTIONLESS : (ell ? EFC_ELLIPTIC : EFC_PYRAMIDAL), c);
          }
        }
      }
      nefc += total;
    }
    overflow = group_max<LPE>(overflow);
    if (overflow) {
      // Row offsets are monotonic, so everything before the first contact that
      // did not fit is contiguous: nefc = end of the last accepted contact (or
      // the limit rows if no contact fit).
      int last = nefc_lim;
      for (int c = lane; c < ncon; c += LPE) if (SI(con_efc)[c] >= 0) {
        const int e = SI(con_efc)[c] + contact_rows(con_dim(c));
        last = e > last ? e : last;
      }
      nefc = group_max<LPE>(last);
      if (lane == 0) SI(imisc)[IM_WARN + DMC_WARN_CNSTRFULL]++;
    }
    // all-dense models: every row belongs to the first (dense) class
    const RowMap rm = {L.d.jfull ? nefc : row_s0, L.d.jfull ? nefc : row_tl0, L.d.jfull ? nefc : nefc_lim};
    if (lane == 0) { SI(imisc)[IM_NEFC] = nefc; SI(imisc)[IM_ROW_S0] = rm.s0; SI(imisc)[IM_ROW_TL0] = rm.tl0; SI(imisc)[IM_ROW_C0] = rm.c0; }
    DMC_WSYNC();
    const auto Jc_base = Jc();
    // contact Jacobian entries: item = (contact, dof); dofs outside the contact's mask are skipped
    for (int idx = lane; idx < ncon*nv; idx += LPE) {
      const int c = idx / nv, dd = idx - c*nv;
      const int r0 = SI(con_efc)[c];
      if (r0 < 0) continue;
      const int slot = L.d.jfull ? dd : mask_slot(con_mask_lo(c), con_mask_hi(c), dd);
      if (slot < 0) continue;
      const int dim = con_dim(c);
      const int b1 = con_b1(c), b2 = con_b2(c);
      const bool in1 = dof_in_chain(MI(body_lastdof)[b1], dd), in2 = dof_in_chain(MI(body_lastdof)[b2], dd);
      T jac[6] = {0, 0, 0, 0, 0, 0};
      {
        const T* cd = S(cdof) + 6*dd; const T* pos = S(con_pos) + 3*c;
        T jp1[3] = {0, 0, 0}, jr1[3] = {0, 0, 0}, jp2[3] = {0, 0, 0}, jr2[3] = {0, 0, 0}, off[3], tmp[3], dp[3], dr[3];
        if (in1) {
          const T* rc = S(subtree_com) + 3*MI(body_rootid)[b1];
          for (int k = 0; k < 3; k++) off[k] = pos[k] - rc[k];
          cross3(tmp, cd, off);
          for (int k = 0; k < 3; k++) { jr1[k] = cd[k]; jp1[k] = cd[3 + k] + tmp[k]; }
        }
        if (in2) {
          const T* rc = S(subtree_com) + 3*MI(body_rootid)[b2];
          for (int k = 0; k < 3; k++) off[k] = pos[k] - rc[k];
          cross3(tmp, cd, off);
          for (int k = 0; k < 3; k++) { jr2[k] = cd[k]; jp2[k] = cd[3 + k] + tmp[k]; }
        }
        for (int k = 0; k < 3; k++) { dp[k] = jp2[k] - jp1[k]; dr[k] = jr2[k] - jr1[k]; }
        const T* fr = conF() + 9*c;
        for (int a = 0; a < 3; a++) { jac[a] = dot3(fr + 3*a, dp); jac[3 + a] = dot3(fr + 3*a, dr); }
      }
      auto put = [&](auto Jw, int K) {      // the contact's rows, entry of this dof (row stride K)
        if (dim == 1) Jw[0] = jac[0];
        else if (L.d.elliptic) for (int k = 0; k < dim; k++) Jw[k*K] = jac[k];
        else for (int k = 1; k < dim; k++) {
          const T f = MR(prm_friction)[3*con_prm(c) + (k < 3 ? 0 : (k == 3 ? 1 : 2))];
          Jw[2*(k - 1)*K] = jac[0] + f*jac[k];
          Jw[(2*(k - 1) + 1)*K] = jac[0] + (-f)*jac[k];
        }
      };
      if (L.d.jfull) put(S(efc_Jd) + r0*nv + dd, nv);      // dofs on neither or both chains come out as exact zeros
      else {
        ((unsigned char*)(SI(con_dofs) + c*L.d.kwords))[slot] = (unsigned char)dd;
        put(Jc_base + (r0 - nefc_lim)*L.d.kmax + slot, L.d.kmax);
      }
    }
    DMC_WSYNC();
    // per-row parameters: D, aref
    for (int i = lane; i < nefc; i += LPE) {
      const T *solref, *solimp; T dA, R;
      const int type = EFC_TYPE(SI(efc_tid)[i]), id = EFC_ID(SI(efc_tid)[i]);
      const T pos = S(efc_aref)[i], margin = S(efc_D)[i];   // staged by the row headers above
      T mu = 0; T dA0 = 0;
      int ell_row = 0; T ell_fj = 0, ell_imp0 = 0;
      if (type == EFC_EQUALITY) {
        solref = MR(eq_solref) + 2*id; solimp = MR(eq_solimp) + 5*id;
        const int et = MI(eq_type)[id], o1 = MI(eq_obj1)[id], o2 = MI(eq_obj2)[id];
        if (et == DMC_EQ_TENDON) dA = MR(tendon_invweight0)[o1];
        else if (et == DMC_EQ_JOINT) {
          dA = MR(dof_invweight0)[MI(jnt_dofadr)[o1]];
          if (o2 >= 0) dA += MR(dof_invweight0)[MI(jnt_dofadr)[o2]];
        } else {
          const int col = (i - MI(eq_rowadr)[id]) < 3 ? 0 : 1;     // translational rows, then (weld) rotational ones
          dA = MR(body_invweight0)[2*o1 + col] + MR(body_invweight0)[2*o2 + col];
        }
      } else if (type == EFC_FRICTION) {
        solref = MR(dof_solref) + 2*id; solimp = MR(dof_solimp) + 5*id;
        dA = MR(dof_invweight0)[id];
      } else if (type == EFC_LIMIT) {
        const int jn = MI(dof_jntid)[id >> 1];
        solref = MRC(jnt_solref) + 2*jn; solimp = MRC(jnt_solimp) + 5*jn;
        dA = MR(dof_invweight0)[id >> 1];
      } else if (L.d.nlimball && type == EFC_TENDON_LIMIT && id >= L.d.ntendon) {      // a ball-joint limit (dense limit row of joint id - ntendon)
        const int jn = id - L.d.ntendon;
        solref = MRC(jnt_solref) + 2*jn; solimp = MRC(jnt_solimp) + 5*jn;
        dA = MR(dof_invweight0)[MI(jnt_dofadr)[jn]];
      } else if (type == EFC_TENDON_LIMIT) {
        solref = MR(tendon_solref_lim) + 2*id; solimp = MR(tendon_solimp_lim) + 5*id;
        dA = MR(tendon_invweight0)[id];
      } else {
        const int b1 = con_b1(id), b2 = con_b2(id), prm = con_prm(id);
        const T tran = MR(body_invweight0)[2*b1] + MR(body_invweight0)[2*b2];
        const T rot = MR(body_invweight0)[2*b1 + 1] + MR(body_invweight0)[2*b2 + 1];
        solref = MR(prm_solref) + 2*prm; solimp = MR(prm_solimp) + 5*prm;
        if (type == EFC_FRICTIONLESS) dA = tran;
        else if (type == EFC_ELLIPTIC) {
          ell_row = i - SI(con_efc)[id];   // 0 normal, 1..2 slide, 3 torsion, 4..5 roll
          dA = ell_row < 3 ? tran : rot;
          if (ell_row > 0) {
            mu = MR(prm_friction)[3*prm]; dA0 = tran;
            ell_fj = MR(prm_friction)[3*prm + (ell_row < 3 ? 0 : (ell_row == 3 ? 1 : 2))];
            ell_imp0 = get_impedance(solimp, S(con_dist)[id], MR(prm_margin)[prm] - MR(prm_gap)[prm]);
          }
        } else {
          const int j = i - SI(con_efc)[id];
          const int k = j/2;   // friction index 0,1: slide; 2: torsion; 3,4: roll
          const T fri = MR(prm_friction)[3*prm + (k < 2 ? 0 : (k == 2 ? 1 : 2))];
          dA = tran + fri*fri*(j < 4 ? tran : rot);
          mu = MR(prm_friction)[3*prm];
          dA0 = tran + mu*mu*tran;
        }
      }
      T ref0 = solref[0], ref1 = solref[1];
      if (!(o.disableflags & DMC_DSBL_REFSAFE) && ref0 > 0) ref0 = t_max(ref0, 2*o.timestep);
      const T imp = get_impedance(solimp, pos, margin);
      // frictional rows: R(first friction) = R(normal)/impratio; regularised mu = mu*sqrt(R1/R0)
      if (type == EFC_PYRAMIDAL) {
        const T R0 = t_max((T)DMC_MINVAL, (1 - imp)*dA0/imp);
        if (o.impratio != 1) { const T R1 = R0 / t_max((T)DMC_MINVAL, o.impratio); mu *= t_sqrt(R1/R0); }
        R = 2*mu*mu*R0;
      } else if (ell_row > 0) {
        const T R0 = t_max((T)DMC_MINVAL, (1 - ell_imp0)*dA0/ell_imp0);
        const T R1 = R0 / t_max((T)DMC_MINVAL, o.impratio);
        R = ell_row == 1 ? R1 : R1*mu*mu/(ell_fj*ell_fj);
      } else R = t_max((T)DMC_MINVAL, (1 - imp)*dA/imp);
      const T dmax = t_max((T)DMC_MINIMP, t_min((T)DMC_MAXIMP, solimp[1]));
      T K, Bd;
      if (ref0 > 0) { K = 1 / t_max((T)DMC_MINVAL, dmax*dmax*ref0*ref0*ref1*ref1); Bd = 2 / t_max((T)DMC_MINVAL, dmax*ref0); }
      else { K = -ref0 / t_max((T)DMC_MINVAL, dmax*dmax); Bd = -ref1 / t_max((T)DMC_MINVAL, dmax); }
      S(efc_D)[i] = 1 / R;
      const T vel = row_dot(i, S(qvel), rm);
      S(efc_aref)[i] = -Bd*vel - K*imp*(pos - margin);
    }
    DMC_WSYNC();
  }

  // ---- velocity stage (mj_comVel, mj_passive, mj_rne) -----------------------------
  DMC_DEV void body_com_vel(int i) {
    T cvel[6], tmp[6], cdd[6];
    for (int a = 0; a < 6; a++) cvel[a] = S(cvel)[6*MI(body_parentid)[i] + a];
    const int bda = MI(body_dofadr)[i];
    int dofs = 0;
    for (int j = MI(body_jntadr)[i]; j < MI(body_jntadr)[i] + MI(body_jntnum)[i]; j++) {
      const int t = MI(jnt_type)[j];
      if (t == DMC_JNT_FREE) {
        for (int k = 0; k < 18; k++) SG(cdof_dot)[6*bda + k] = 0;
        for (int k = 0; k < 3; k++) for (int a = 0; a < 6; a++) cvel[a] += S(cdof)[6*(bda + k) + a] * S(qvel)[bda + k];
        dofs += 3;
      }
      if (t == DMC_JNT_FREE || t == DMC_JNT_BALL) {
        for (int k = 0; k < 3; k++) {
          cross_motion(cdd, cvel, S(cdof) + 6*(bda + dofs + k));
          for (int a = 0; a < 6; a++) SG(cdof_dot)[6*(bda + dofs + k) + a] = cdd[a];
        }
        for (int a = 0; a < 6; a++) tmp[a] = 0;
        for (int k = 0; k < 3; k++) for (int a = 0; a < 6; a++) tmp[a] += S(cdof)[6*(bda + dofs + k) + a] * S(qvel)[bda + dofs + k];
        for (int a = 0; a < 6; a++) cvel[a] += tmp[a];
        dofs += 3;
      } else {
        cross_motion(cdd, cvel, S(cdof) + 6*(bda + dofs));
        for (int a = 0; a < 6; a++) SG(cdof_dot)[6*(bda + dofs) + a] = cdd[a];
        for (int a = 0; a < 6; a++) cvel[a] += S(cdof)[6*(bda + dofs) + a] * S(qvel)[bda + dofs];
        dofs += 1;
      }
    }
    for (int a = 0; a < 6; a++) S(cvel)[6*i + a] = cvel[a];
  }
  DMC_DEV void com_vel() {
    if (L.d.nv <= 64) {
      // mj_comVel without a pass per tree level: the velocity a dof sees (cdof_dot = cvel x cdof) is the sum of
      // cdof * qvel over the dofs BEFORE it on its path -- before its whole triple for the rotational dofs of a ball /
      // free joint, nothing for a free joint's translations -- and a body's velocity the sum over its whole path (the
      // ancestor mask of its last dof, ascending = the order the body-by-body recursion adds them in).
      const int nv = L.d.nv;
      FOR_LANES(k, nv) {
        const int j = MI(dof_jntid)[k], t = MI(jnt_type)[j], da = MI(jnt_dofadr)[j];
        T cdd[6] = {0, 0, 0, 0, 0, 0};
        if (!(t == DMC_JNT_FREE && k < da + 3)) {
          const int start = t == DMC_JNT_FREE ? da + 3 : (t == DMC_JNT_BALL ? da : k);
          unsigned lo = (unsigned)MI(dof_anc_lo)[k], hi = nv > 32 ? (unsigned)MI(dof_anc_hi)[k] : 0u;
          if (start < 32) { lo &= (1u << start) - 1u; hi = 0; } else if (start < 64) hi &= (1u << (start - 32)) - 1u;
          T cv[6] = {0, 0, 0, 0, 0, 0};
          while (lo) { const int m = __builtin_ctz(lo); lo &= lo - 1; const T v = S(qvel)[m]; const T* cd = S(cdof) + 6*m; for (int a = 0; a < 6; a++) cv[a] += cd[a]*v; }
          while (hi) { const int m = 32 + __builtin_ctz(hi); hi &= hi - 1; const T v = S(qvel)[m]; const T* cd = S(cdof) + 6*m; for (int a = 0; a < 6; a++) cv[a] += cd[a]*v; }
          cross_motion(cdd, cv, S(cdof) + 6*k);
        }
        for (int a = 0; a < 6; a++) SG(cdof_dot)[6*k + a] = cdd[a];
      }
      FOR_LANES(i, L.d.nbody) {
        if (i == 0) continue;
        T cv[6] = {0, 0, 0, 0, 0, 0};
        const int ld = MI(body_lastdof)[i];
        if (ld >= 0) {
          unsigned lo = (unsigned)MI(dof_anc_lo)[ld], hi = nv > 32 ? (unsigned)MI(dof_anc_hi)[ld] : 0u;
          while (lo) { const int m = __builtin_ctz(lo); lo &= lo - 1; const T v = S(qvel)[m]; const T* cd = S(cdof) + 6*m; for (int a = 0; a < 6; a++) cv[a] += cd[a]*v; }
          while (hi) { const int m = 32 + __builtin_ctz(hi); hi &= hi - 1; const T v = S(qvel)[m]; const T* cd = S(cdof) + 6*m; for (int a = 0; a < 6; a++) cv[a] += cd[a]*v; }
        }
        for (int a = 0; a < 6; a++) S(cvel)[6*i + a] = cv[a];
      }
      DMC_WSYNC();
      return;
    }
    for (int lev = 0; lev < L.d.nlevel; lev++) {
      const int a0 = MI(level_adr)[lev], a1 = MI(level_adr)[lev + 1];
      for (int k = a0 + lane; k < a1; k += LPE) body_com_vel(MI(level_body)[k]);
      DMC_WSYNC();
    }
  }
  // mj_passive fluid forces, inertia-box model: each body with mass is replaced by the box of
  // equal inertia; Stokes (viscosity) and quadratic (density) drag on its local inertial-frame
  // velocity; the wrench acts at the body COM.  Per-body wrenches are parked in cfrc_ext (free
  // until mj_rnePostConstraint), then projected on the dofs.
  DMC_DEV void fluid_forces() {
    const int nb = L.d.nbody, nv = L.d.nv;
    FOR_LANES(b, nb) {
      T wr[6] = {0, 0, 0, 0, 0, 0};
      const T mass = MR(body_mass)[b];
      if (b > 0 && mass >= (T)DMC_MINVAL) {
        const T* I = MRC(body_inertia) + 3*b;
        const T box[3] = {t_sqrt(t_max((T)DMC_MINVAL, I[1] + I[2] - I[0]) / mass * (T)6),
                          t_sqrt(t_max((T)DMC_MINVAL, I[0] + I[2] - I[1]) / mass * (T)6),
                          t_sqrt(t_max((T)DMC_MINVAL, I[0] + I[1] - I[2]) / mass * (T)6)};
        T q[4], im[9], lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0};
        mul_quat(q, SG(xquat) + 4*b, MRC(body_iquat) + 4*b);
        quat2mat(im, q);
        object_velocity(b, S(xipos) + 3*b, im, lvel);
        const T pi = (T)3.14159265358979323846;
        if (o.viscosity > 0) {
          const T diam = (box[0] + box[1] + box[2]) / (T)3;
          for (int k = 0; k < 3; k++) { lfrc[k] = -pi*diam*diam*diam*o.viscosity * lvel[k]; lfrc[3 + k] = -(T)3*pi*diam*o.viscosity * lvel[3 + k]; }
        }
        if (o.density > 0) {
          const T rho = o.density;
          lfrc[3] -= (T)0.5*rho*box[1]*box[2]*t_abs(lvel[3])*lvel[3];
          lfrc[4] -= (T)0.5*rho*box[0]*box[2]*t_abs(lvel[4])*lvel[4];
          lfrc[5] -= (T)0.5*rho*box[0]*box[1]*t_abs(lvel[5])*lvel[5];
          lfrc[0] -= rho*box[0]*(box[1]*box[1]*box[1]*box[1] + box[2]*box[2]*box[2]*box[2])*t_abs(lvel[0])*lvel[0]/(T)64;
          lfrc[1] -= rho*box[1]*(box[0]*box[0]*box[0]*box[0] + box[2]*box[2]*box[2]*box[2])*t_abs(lvel[1])*lvel[1]/(T)64;
          lfrc[2] -= rho*box[2]*(box[0]*box[0]*box[0]*box[0] + box[1]*box[1]*box[1]*box[1])*t_abs(lvel[2])*lvel[2]/(T)64;
        }
        mul_mat_vec3(wr, im, lfrc); mul_mat_vec3(wr + 3, im, lfrc + 3);   // [torque, force] in the world frame
      }
      for (int k = 0; k < 6; k++) S(cfrc_ext)[6*b + k] = wr[k];
    }
    DMC_WSYNC();
    FOR_LANES(dd, nv) {
      T f = 0;
      const T* cd = S(cdof) + 6*dd;
      for (int b = 1; b < nb; b++) {
        if (!dof_in_chain(MI(body_lastdof)[b], dd)) continue;
        const T* rc = S(subtree_com) + 3*MI(body_rootid)[b];
        T off[3] = {S(xipos)[3*b] - rc[0], S(xipos)[3*b + 1] - rc[1], S(xipos)[3*b + 2] - rc[2]}, tmp[3];
        cross3(tmp, cd, off);
        const T* wr = S(cfrc_ext) + 6*b;
        f += (cd[3] + tmp[0])*wr[3] + (cd[4] + tmp[1])*wr[4] + (cd[5] + tmp[2])*wr[5] + cd[0]*wr[0] + cd[1]*wr[1] + cd[2]*wr[2];
      }
      S(qfrc_passive)[dd] += f;
    }
    DMC_WSYNC();
  }
  DMC_DEV void passive_and_rne() {
    const int nv = L.d.nv;
    FOR_LANES(i, nv) {
      T f = 0;
      const int j = MI(dof_jntid)[i], t = MI(jnt_type)[j];
      const T k = MRC(jnt_stiffness)[j];
      if (!(o.disableflags & DMC_DSBL_SPRING) && k != 0 && (t == DMC_JNT_SLIDE || t == DMC_JNT_HINGE)) {
        const int qa = MI(jnt_qposadr)[j];
        f -= k * (S(qpos)[qa] - MR(qpos_spring)[qa]);
      }
      if (!(o.disableflags & DMC_DSBL_DAMPER)) f -= MR(dof_damping)[i] * S(qvel)[i];
      // fixed-tendon springs / dampers (each lane re-derives the few tendon lengths it needs)
      for (int t = 0; t < L.d.ntendon; t++) {
        const T kt = MR(tendon_stiffness)[t], bt = MR(tendon_damping)[t];
        if (kt == 0 && bt == 0) continue;
        T len = 0, vel = 0, mine = 0;
        for (int w = MI(tendon_adr)[t]; w < MI(tendon_adr)[t] + MI(tendon_num)[t]; w++) {
          len += MR(wrap_prm)[w] * S(qpos)[MI(wrap_qpos)[w]]; vel += MR(wrap_prm)[w] * S(qvel)[MI(wrap_dof)[w]];
          if (MI(wrap_dof)[w] == i) mine += MR(wrap_prm)[w];
        }
        T ft = 0;
        if (kt != 0 && !(o.disableflags & DMC_DSBL_SPRING)) ft -= kt * (len - MR(tendon_lengthspring)[t]);
        if (bt != 0 && !(o.disableflags & DMC_DSBL_DAMPER)) ft -= bt * vel;
        f += mine * ft;
      }
      S(qfrc_passive)[i] = f;
    }
    if (L.d.fluid) fluid_forces();
    if (lane == 0) {
      T* ca = S(cacc);
      ca[0] = ca[1] = ca[2] = 0; ca[3] = ca[4] = ca[5] = 0;
      if (!(o.disableflags & DMC_DSBL_GRAVITY)) { ca[3] = -o.gravity[0]; ca[4] = -o.gravity[1]; ca[5] = -o.gravity[2]; }
    }
    DMC_WSYNC();
    if (L.d.dfs && L.d.nv <= 64) {
      // RNE without a pass per tree level: a body's acceleration is the sum of cdof_dot * qvel over the dofs on its path
      // (the ancestor mask of its last dof, ascending = root first), so every body evaluates its own chain and its
      // inertial force in ONE pass; the bias force of dof i is cdof_i . (sum of those forces over the subtree of its
      // body = an id range, depth-first numbering), evaluated by the dof's lane in a second one.
      FOR_LANES(i, L.d.nbody) {
        if (i == 0) continue;
        T ca[6], cf[6], tmp[6], tmp1[6];
        for (int a = 0; a < 6; a++) ca[a] = S(cacc)[a];
        const int ld = MI(body_lastdof)[i];
        if (ld >= 0) {
          unsigned lo = (unsigned)MI(dof_anc_lo)[ld], hi = nv > 32 ? (unsigned)MI(dof_anc_hi)[ld] : 0u;
          while (lo) { const int k = __builtin_ctz(lo); lo &= lo - 1; const T v = S(qvel)[k]; const T* cd = SG(cdof_dot) + 6*k; for (int a = 0; a < 6; a++) ca[a] += cd[a]*v; }
          while (hi) { const int k = 32 + __builtin_ctz(hi); hi &= hi - 1; const T v = S(qvel)[k]; const T* cd = SG(cdof_dot) + 6*k; for (int a = 0; a < 6; a++) ca[a] += cd[a]*v; }
        }
        for (int a = 0; a < 6; a++) S(cacc)[6*i + a] = ca[a];
        mul_inert_vec(cf, SG(cinert) + 10*i, ca);
        mul_inert_vec(tmp, SG(cinert) + 10*i, S(cvel) + 6*i);
        cross_force(tmp1, S(cvel) + 6*i, tmp);
        for (int a = 0; a < 6; a++) S(cfrc)[6*i + a] = cf[a] + tmp1[a];
      }
      DMC_WSYNC();
      FOR_LANES(i, nv) {
        const int b = MI(dof_bodyid)[i], e = MI(body_subend)[b];
        T f[6] = {0, 0, 0, 0, 0, 0};
        for (int c = b; c < e; c++) for (int a = 0; a < 6; a++) f[a] += S(cfrc)[6*c + a];
        S(qfrc_bias)[i] = dot_n(S(cdof) + 6*i, f, 6);
      }
      DMC_WSYNC();
      return;
    }
    for (int lev = 0; lev < L.d.nlevel; lev++) {
      const int a0 = MI(level_adr)[lev], a1 = MI(level_adr)[lev + 1];
      for (int kk = a0 + lane; kk < a1; kk += LPE) {
        const int i = MI(level_body)[kk], bda = MI(body_dofadr)[i];
        T tmp[6], tmp1[6], ca[6], cf[6];
        for (int a = 0; a < 6; a++) tmp[a] = 0;
        for (int k = 0; k < MI(body_dofnum)[i]; k++) for (int a = 0; a < 6; a++) tmp[a] += SG(cdof_dot)[6*(bda + k) + a] * S(qvel)[bda + k];
        for (int a = 0; a < 6; a++) ca[a] = S(cacc)[6*MI(body_parentid)[i] + a] + tmp[a];
        for (int a = 0; a < 6; a++) S(cacc)[6*i + a] = ca[a];
        mul_inert_vec(cf, SG(cinert) + 10*i, ca);
        mul_inert_vec(tmp, SG(cinert) + 10*i, S(cvel) + 6*i);
        cross_force(tmp1, S(cvel) + 6*i, tmp);
        for (int a = 0; a < 6; a++) S(cfrc)[6*i + a] = cf[a] + tmp1[a];
      }
      DMC_WSYNC();
    }
    for (int lev = L.d.nlevel - 2; lev >= 0; lev--) {
      const int a0 = MI(level_adr)[lev], cnt = MI(level_adr)[lev + 1] - a0;
      for (int idx = lane; idx < cnt*6; idx += LPE) {
        const int b = MI(level_body)[a0 + idx/6], comp = idx % 6;
        const int c0 = MI(child_adr)[b], c1 = MI(child_adr)[b + 1];
        if (c1 > c0) {
          T v = S(cfrc)[6*b + comp];
          for (int c = c0; c < c1; c++) v += S(cfrc)[6*MI(child_list)[c] + comp];
          S(cfrc)[6*b + comp] = v;
        }
      }
      DMC_WSYNC();
    }
    FOR_LANES(i, nv) S(qfrc_bias)[i] = dot_n(S(cdof) + 6*i, S(cfrc) + 6*MI(dof_bodyid)[i], 6);
    DMC_WSYNC();
  }

  // ---- sensors (position / velocity stage) -------------------------------------------
  // subtreelinvel sensors (mj_subtreeVel restricted to what is read): linear momentum of
  // the bodies below `root` about their own COMs, summed by one reduction, over the subtree mass
  DMC_DEV void subtree_linvel_sensors() {
    for (int k = 0; k < L.d.nstv; k++) {
      const int sidx = MI(stv_sensor)[k], root = MI(sensor_objid)[sidx];
      T mom[3] = {0, 0, 0};
      FOR_LANES(i, L.d.nbody) {
        const unsigned m = (unsigned)(root < 32 ? MI(body_anc_lo)[i] : MI(body_anc_hi)[i]);
        if (!((m >> (root & 31)) & 1u)) continue;
        const T* rc = S(subtree_com) + 3*MI(body_rootid)[i];
        T dif[3] = {S(xipos)[3*i] - rc[0], S(xipos)[3*i + 1] - rc[1], S(xipos)[3*i + 2] - rc[2]}, tmp[3];
        cross3(tmp, dif, S(cvel) + 6*i);
        for (int c = 0; c < 3; c++) mom[c] += MR(body_mass)[i] * (S(cvel)[6*i + 3 + c] - tmp[c]);
      }
      const T inv = 1 / t_max((T)DMC_MINVAL, MR(body_subtreemass)[root]);
      for (int c = 0; c < 3; c++) { const T v = group_sum<LPE>(mom[c]) * inv; if (lane == 0) S(sensordata)[MI(sensor_adr)[sidx] + c] = v; }
    }
  }
  DMC_DEV void object_velocity(int body, const T* pos, const T* mat, T* res) {
    const T* rc = S(subtree_com) + 3*MI(body_rootid)[body];
    T dif[3] = {pos[0] - rc[0], pos[1] - rc[1], pos[2] - rc[2]}, tmp[3];
    const T* cv = S(cvel) + 6*body;
    cross3(tmp, dif, cv);
    T lin[3] = {cv[3] - tmp[0], cv[4] - tmp[1], cv[5] - tmp[2]};
    mul_matT_vec3(res, mat, cv); mul_matT_vec3(res + 3, mat, lin);
  }
  // ---- acceleration-stage support: contact wrench, mj_rnePostConstraint, touch rays --------
  // contact force in the contact frame [normal, tangent1, tangent2, torsion, roll1, roll2]
  DMC_DEV void contact_force_local(int c, T* f6) {
    for (int k = 0; k < 6; k++) f6[k] = 0;
    const int r0 = SI(con_efc)[c];
    if (r0 < 0) return;
    const int dim = con_dim(c);
    const T* f = S(efc_force) + r0;
    if (dim == 1) { f6[0] = f[0]; return; }
    if (L.d.elliptic) { for (int k = 0; k < dim; k++) f6[k] = f[k]; return; }
    for (int k = 0; k < 2*(dim - 1); k++) f6[0] += f[k];
    for (int k = 1; k < dim; k++) f6[k] = (f[2*(k - 1)] - f[2*(k - 1) + 1]) * MR(prm_friction)[3*con_prm(c) + (k < 3 ? 0 : (k == 3 ? 1 : 2))];
  }
  DMC_DEV void rne_post_constraint() {
    const int nb = L.d.nbody, ncon = SI(imisc)[IM_NCON];
    // external (contact) wrench per body, contacts visited in index order
    FOR_LANES(b, nb) {
      T acc[6] = {0, 0, 0, 0, 0, 0};
      if (b > 0) for (int c = 0; c < ncon; c++) {
        if (SI(con_efc)[c] < 0) continue;
        const int b1 = con_b1(c), b2 = con_b2(c);
        if (b1 != b && b2 != b) continue;
        T lf[6], gf[3], gt[3], dif[3], t[3];
        contact_force_local(c, lf);
        mul_matT_vec3(gf, conF() + 9*c, lf); mul_matT_vec3(gt, conF() + 9*c, lf + 3);
        const T* rc = S(subtree_com) + 3*MI(body_rootid)[b];
        for (int k = 0; k < 3; k++) dif[k] = S(con_pos)[3*c + k] - rc[k];
        cross3(t, dif, gf);
        if (b1 == b) for (int k = 0; k < 3; k++) { acc[k] += -(gt[k] + t[k]); acc[3 + k] += -gf[k]; }
        if (b2 == b) for (int k = 0; k < 3; k++) { acc[k] += (gt[k] + t[k]); acc[3 + k] += gf[k]; }
      }
      if (b > 0 && o.xfrc) {      // applied Cartesian wrench, about the root's subtree COM like the contact wrenches
        T xf[6], dif[3], t[3]; load_xfrc(b, xf);
        const T* rc = S(subtree_com) + 3*MI(body_rootid)[b];
        for (int k = 0; k < 3; k++) dif[k] = S(xipos)[3*b + k] - rc[k];
        cross3(t, dif, xf);
        for (int k = 0; k < 3; k++) { acc[k] += xf[3 + k] + t[k]; acc[3 + k] += xf[k]; }
      }
      for (int k = 0; k < 6; k++) S(cfrc_ext)[6*b + k] = acc[k];
    }
    if (lane == 0) {
      T* ca = S(cacc);
      ca[0] = ca[1] = ca[2] = 0; ca[3] = ca[4] = ca[5] = 0;
      if (!(o.disableflags & DMC_DSBL_GRAVITY)) { ca[3] = -o.gravity[0]; ca[4] = -o.gravity[1]; ca[5] = -o.gravity[2]; }
      for (int k = 0; k < 6; k++) S(cfrc)[k] = 0;
    }
    DMC_WSYNC();
    for (int lev = 0; lev < L.d.nlevel; lev++) {
      const int a0 = MI(level_adr)[lev], a1 = MI(level_adr)[lev + 1];
      for (int kk = a0 + lane; kk < a1; kk += LPE) {
        const int i = MI(level_body)[kk], bda = MI(body_dofadr)[i];
        T csum[6], tmp[6], tmp1[6], tmp2[6];
        for (int a = 0; a < 6; a++) csum[a] = S(cacc)[6*MI(body_parentid)[i] + a];
        for (int k = 0; k < MI(body_dofnum)[i]; k++) for (int a = 0; a < 6; a++)
          csum[a] += SG(cdof_dot)[6*(bda + k) + a]*S(qvel)[bda + k] + S(cdof)[6*(bda + k) + a]*S(qacc)[bda + k];
        for (int a = 0; a < 6; a++) S(cacc)[6*i + a] = csum[a];
        mul_inert_vec(tmp, SG(cinert) + 10*i, csum);
        mul_inert_vec(tmp1, SG(cinert) + 10*i, S(cvel) + 6*i);
        cross_force(tmp2, S(cvel) + 6*i, tmp1);
        for (int a = 0; a < 6; a++) S(cfrc)[6*i + a] = tmp[a] + tmp2[a] - S(cfrc_ext)[6*i + a];
      }
      DMC_WSYNC();
    }
    for (int lev = L.d.nlevel - 2; lev >= 0; lev--) {
      const int a0 = MI(level_adr)[lev], cnt = MI(level_adr)[lev + 1] - a0;
      for (int idx = lane; idx < cnt*6; idx += LPE) {
        const int b = MI(level_body)[a0 + idx/6], comp = idx % 6;
        const int c0 = MI(child_adr)[b], c1 = MI(child_adr)[b + 1];
        if (c1 > c0) {
          T v = S(cfrc)[6*b + comp];
          for (int c = c0; c < c1; c++) v += S(cfrc)[6*MI(child_list)[c] + comp];
          S(cfrc)[6*b + comp] = v;
        }
      }
      DMC_WSYNC();
    }
  }
  // ray (pnt, vec) against a site volume in its own frame; distance or -1
  DMC_DEV static T ray_geom(const T* pos, const T* mat, const T* size, const T* pnt, const T* vec, int type) {
    T dif[3] = {pnt[0] - pos[0], pnt[1] - pos[1], pnt[2] - pos[2]}, lp[3], lv[3];
    mul_matT_vec3(lp, mat, dif); mul_matT_vec3(lv, mat, vec);
    T best = -1;
    if (type == DMC_GEOM_SPHERE || type == DMC_GEOM_CAPSULE) {
      const T r = size[0];
      const int nparts = type == DMC_GEOM_CAPSULE ? 3 : 1;
      for (int part = 0; part < nparts; part++) {
        T a, b, c;
        if (type == DMC_GEOM_CAPSULE && part == 0) {
          a = lv[0]*lv[0] + lv[1]*lv[1]; b = lp[0]*lv[0] + lp[1]*lv[1]; c = lp[0]*lp[0] + lp[1]*lp[1] - r*r;
        } else {
          const T cz = type == DMC_GEOM_CAPSULE ? (part == 1 ? size[1] : -size[1]) : (T)0;
          T q[3] = {lp[0], lp[1], lp[2] - cz};
          a = dot3(lv, lv); b = dot3(q, lv); c = dot3(q, q) - r*r;
        }
        if (a < (T)DMC_MINVAL) continue;
        const T det = b*b - a*c;
        if (det < 0) continue;
        const T sq = t_sqrt(det);
        for (int k = 0; k < 2; k++) {
          const T x = k == 0 ? (-b - sq)/a : (-b + sq)/a;
          if (x < 0) continue;
          const T z = lp[2] + x*lv[2];
          if (type == DMC_GEOM_CAPSULE) {
            if (part == 0 && t_abs(z) > size[1]) continue;
            if (part == 1 && z < size[1]) continue;
            if (part == 2 && z > -size[1]) continue;
          }
          if (best < 0 || x < best) best = x;
        }
      }
      return best;
    }
    if (type == DMC_GEOM_ELLIPSOID) {
      T q[3] = {lp[0]/size[0], lp[1]/size[1], lp[2]/size[2]}, w[3] = {lv[0]/size[0], lv[1]/size[1], lv[2]/size[2]};
      const T a = dot3(w, w), b = dot3(q, w), c = dot3(q, q) - 1;
      if (a < (T)DMC_MINVAL) return -1;
      const T det = b*b - a*c;
      if (det < 0) return -1;
      const T sq = t_sqrt(det), x0 = (-b - sq)/a, x1 = (-b + sq)/a;
      return x0 >= 0 ? x0 : (x1 >= 0 ? x1 : (T)-1);
    }
    if (type == DMC_GEOM_BOX) {
      if (t_abs(lp[0]) <= size[0] && t_abs(lp[1]) <= size[1] && t_abs(lp[2]) <= size[2]) return 0;
      for (int ax = 0; ax < 3; ax++) {
        if (t_abs(lv[ax]) < (T)DMC_MINVAL) continue;
        for (int sg = -1; sg <= 1; sg += 2) {
          const T x = (sg*size[ax] - lp[ax]) / lv[ax];
          if (x < 0) continue;
          const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
          if (t_abs(lp[a1] + x*lv[a1]) <= size[a1] && t_abs(lp[a2] + x*lv[a2]) <= size[a2])
            if (best < 0 || x < best) best = x;
        }
      }
      return best;
    }
    return -1;
  }
  // rays of rangefinder sensors: every geom type (planes are front-side only and finite where their
  // half-sizes are positive; a ray that starts inside a box leaves through a face)
  DMC_DEV static T ray_geom_any(const T* pos, const T* mat, const T* size, const T* pnt, const T* vec, int type) {
    if (type != DMC_GEOM_PLANE && type != DMC_GEOM_CYLINDER && type != DMC_GEOM_BOX) return ray_geom(pos, mat, size, pnt, vec, type);
    T dif[3] = {pnt[0] - pos[0], pnt[1] - pos[1], pnt[2] - pos[2]}, lp[3], lv[3];
    mul_matT_vec3(lp, mat, dif); mul_matT_vec3(lv, mat, vec);
    T best = -1;
    if (type == DMC_GEOM_PLANE) {
      if (lv[2] > -(T)DMC_MINVAL) return -1;
      const T x = -lp[2]/lv[2];
      if (x < 0) return -1;
      const T px = lp[0] + x*lv[0], py = lp[1] + x*lv[1];
      if ((size[0] <= 0 || t_abs(px) <= size[0]) && (size[1] <= 0 || t_abs(py) <= size[1])) return x;
      return -1;
    }
    if (type == DMC_GEOM_CYLINDER) {
      const T a = lv[0]*lv[0] + lv[1]*lv[1], b = lp[0]*lv[0] + lp[1]*lv[1], c = lp[0]*lp[0] + lp[1]*lp[1] - size[0]*size[0];
      if (a >= (T)DMC_MINVAL) {
        const T det = b*b - a*c;
        if (det >= 0) {
          const T sq = t_sqrt(det);
          for (int k = 0; k < 2; k++) {
            const T x = k == 0 ? (-b - sq)/a : (-b + sq)/a;
            if (x >= 0 && t_abs(lp[2] + x*lv[2]) <= size[1]) if (best < 0 || x < best) best = x;
          }
        }
      }
      if (t_abs(lv[2]) >= (T)DMC_MINVAL) for (int sg = -1; sg <= 1; sg += 2) {
        const T x = (sg*size[1] - lp[2]) / lv[2];
        if (x < 0) continue;
        const T px = lp[0] + x*lv[0], py = lp[1] + x*lv[1];
        if (px*px + py*py <= size[0]*size[0]) if (best < 0 || x < best) best = x;
      }
      return best;
    }
    for (int ax = 0; ax < 3; ax++) {
      if (t_abs(lv[ax]) < (T)DMC_MINVAL) continue;
      for (int sg = -1; sg <= 1; sg += 2) {
        const T x = (sg*size[ax] - lp[ax]) / lv[ax];
        if (x < 0) continue;
        const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
        if (t_abs(lp[a1] + x*lv[a1]) <= size[a1] && t_abs(lp[a2] + x*lv[a2]) <= size[a2])
          if (best < 0 || x < best) best = x;
      }
    }
    return best;
  }
  DMC_DEV void sensors_acc() {
    if ((o.disableflags & DMC_DSBL_SENSOR) || L.d.nsensor == 0) return;
    int need = 0;
    for (int i = 0; i < L.d.nsensor; i++) {
      const int t = MI(sensor_type)[i];
      if (MI(sensor_stage)[i] == DMC_STAGE_ACC && (t == DMC_SENS_ACCELEROMETER || t == DMC_SENS_FORCE || t == DMC_SENS_TORQUE)) need = 1;
    }
    if (need) rne_post_constraint();
    const int ncon = SI(imisc)[IM_NCON];
    FOR_LANES(i, L.d.nsensor) {
      if (MI(sensor_stage)[i] != DMC_STAGE_ACC) continue;
      T* out = S(sensordata) + MI(sensor_adr)[i];
      const int id = MI(sensor_objid)[i], t = MI(sensor_type)[i];
      if (t == DMC_SENS_ACTUATORFRC) { out[0] = S(actuator_force)[id]; continue; }
      const int body = MI(site_bodyid)[id];
      T v[3], q[4], smat[9], spos[3];
      mul_mat_vec3(v, S(xmat) + 9*body, MRS(site_pos) + 3*id);
      for (int k = 0; k < 3; k++) spos[k] = S(xpos)[3*body + k] + v[k];
      mul_quat(q, SG(xquat) + 4*body, MRS(site_quat) + 4*id);
      quat2mat(smat, q);
      const T* rc = S(subtree_com) + 3*MI(body_rootid)[body];
      if (t == DMC_SENS_TOUCH) {
        T tot = 0;
        for (int c = 0; c < ncon; c++) {
          if (SI(con_efc)[c] < 0) continue;
          const int b1 = con_b1(c), b2 = con_b2(c);
          if (b1 != body && b2 != body) continue;
          T lf[6]; contact_force_local(c, lf);
          if (lf[0] <= 0) continue;
          const T* fr = conF() + 9*c;
          T ray[3] = {fr[0]*lf[0], fr[1]*lf[0], fr[2]*lf[0]};
          normalize3(ray);
          if (b2 == body) { ray[0] = -ray[0]; ray[1] = -ray[1]; ray[2] = -ray[2]; }
          if (ray_geom(spos, smat, MRS(site_size) + 3*id, S(con_pos) + 3*c, ray, MI(site_type)[id]) >= 0) tot += lf[0];
        }
        out[0] = tot;
      } else if (t == DMC_SENS_ACCELEROMETER) {
        T dif[3] = {spos[0] - rc[0], spos[1] - rc[1], spos[2] - rc[2]}, tmp[3], lin[3], cor[3], tmp2[3], vlin[3];
        const T* ca = S(cacc) + 6*body; const T* cv = S(cvel) + 6*body;
        cross3(tmp, dif, ca);
        for (int k = 0; k < 3; k++) lin[k] = ca[3 + k] - tmp[k];
        cross3(tmp2, dif, cv);
        for (int k = 0; k < 3; k++) vlin[k] = cv[3 + k] - tmp2[k];
        cross3(cor, cv, vlin);
        for (int k = 0; k < 3; k++) lin[k] += cor[k];
        mul_matT_vec3(out, smat, lin);
      } else if (t == DMC_SENS_FORCE) {
        mul_matT_vec3(out, smat, S(cfrc) + 6*body + 3);
      } else if (t == DMC_SENS_TORQUE) {
        T dif[3] = {spos[0] - rc[0], spos[1] - rc[1], spos[2] - rc[2]}, tmp[3], tq[3];
        cross3(tmp, dif, S(cfrc) + 6*body + 3);
        for (int k = 0; k < 3; k++) tq[k] = S(cfrc)[6*body + k] - tmp[k];
        mul_matT_vec3(out, smat, tq);
      }
    }
    DMC_WSYNC();
  }
  // pose (origin, quaternion) of a sensor's reference frame (mj_sensorPos)
  DMC_DEV void sensor_ref_pose(int rt, int rid, T* p, T* q) {
    if (rt == DMC_OBJ_SITE) {
      const int b = MI(site_bodyid)[rid]; T v[3];
      mul_mat_vec3(v, S(xmat) + 9*b, MRS(site_pos) + 3*rid);
      for (int k = 0; k < 3; k++) p[k] = S(xpos)[3*b + k] + v[k];
      mul_quat(q, SG(xquat) + 4*b, MRS(site_quat) + 4*rid);
    } else if (rt == DMC_OBJ_GEOM) {
      for (int k = 0; k < 3; k++) p[k] = S(geom_xpos)[3*rid + k];
      mul_quat(q, SG(xquat) + 4*MI(geom_bodyid)[rid], MRC(geom_quat) + 4*rid);
    } else if (rt == DMC_OBJ_BODY) {
      for (int k = 0; k < 3; k++) p[k] = S(xipos)[3*rid + k];
      mul_quat(q, SG(xquat) + 4*rid, MRC(body_iquat) + 4*rid);
    } else {
      for (int k = 0; k < 3; k++) p[k] = S(xpos)[3*rid + k];
      for (int k = 0; k < 4; k++) q[k] = SG(xquat)[4*rid + k];
    }
  }
  DMC_DEV void sensors(int stage) {
    if ((o.disableflags & DMC_DSBL_SENSOR) || L.d.nsensor == 0) return;
    if (L.d.nstv && stage == DMC_STAGE_VEL) subtree_linvel_sensors();
    FOR_LANES(i, L.d.nsensor) {
      if (MI(sensor_stage)[i] != stage) continue;
      T* out = S(sensordata) + MI(sensor_adr)[i];
      const int id = MI(sensor_objid)[i], t = MI(sensor_type)[i];
      if (t == DMC_SENS_JOINTPOS) out[0] = S(qpos)[MI(jnt_qposadr)[id]];
      else if (t == DMC_SENS_JOINTVEL) out[0] = S(qvel)[MI(jnt_dofadr)[id]];
      else if (t == DMC_SENS_ACTUATORFRC) out[0] = S(actuator_force)[id];
      else if (t == DMC_SENS_SUBTREECOM) for (int k = 0; k < 3; k++) out[k] = S(subtree_com)[3*id + k];
      else if (t >= DMC_SENS_FRAMEXAXIS && t <= DMC_SENS_FRAMEZAXIS) {
        const int otr = MI(sensor_objtype)[i], ot = otr & 255, c = t - DMC_SENS_FRAMEXAXIS;
        if (ot == DMC_OBJ_SITE || ot == DMC_OBJ_BODY) {
          // column c of quat2mat(q) = q rotating the unit vector e_c (no matrix held in memory:
          // selecting between a local array and an LDS array through a pointer pins it in scratch)
          T q[4], e[3] = {c == 0 ? (T)1 : (T)0, c == 1 ? (T)1 : (T)0, c == 2 ? (T)1 : (T)0}, col[3];
          if (ot == DMC_OBJ_SITE) mul_quat(q, SG(xquat) + 4*MI(site_bodyid)[id], MRS(site_quat) + 4*id);
          else mul_quat(q, SG(xquat) + 4*id, MRC(body_iquat) + 4*id);
          rot_vec_quat(col, e, q);
          out[0] = col[0]; out[1] = col[1]; out[2] = col[2];
        } else {
          const T* Rp = ot == DMC_OBJ_GEOM ? SG(geom_xmat) + 9*id : S(xmat) + 9*id;
          out[0] = Rp[c]; out[1] = Rp[3 + c]; out[2] = Rp[6 + c];
        }
        if (otr >> 16) {      // the axis in the reference frame: conj(q_ref) rotates it
          T qr[4], pr[3], ax[3] = {out[0], out[1], out[2]}, r[3];
          sensor_ref_pose((otr >> 8) & 255, (otr >> 16) - 1, pr, qr);
          qr[1] = -qr[1]; qr[2] = -qr[2]; qr[3] = -qr[3];
          rot_vec_quat(r, ax, qr);
          out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
        }
      }
      else if (t == DMC_SENS_FRAMEPOS) {
        const int otr = MI(sensor_objtype)[i], ot = otr & 255;
        if (ot == DMC_OBJ_SITE) {
          const int b = MI(site_bodyid)[id]; T v[3];
          mul_mat_vec3(v, S(xmat) + 9*b, MRS(site_pos) + 3*id);
          for (int k = 0; k < 3; k++) out[k] = S(xpos)[3*b + k] + v[k];
        } else {
          const T* p = ot == DMC_OBJ_GEOM ? S(geom_xpos) + 3*id : (ot == DMC_OBJ_BODY ? S(xipos) + 3*id : S(xpos) + 3*id);
          for (int k = 0; k < 3; k++) out[k] = p[k];
        }
        if (otr >> 16) {      // R_ref' (p - p_ref)
          T qr[4], pr[3], r[3];
          sensor_ref_pose((otr >> 8) & 255, (otr >> 16) - 1, pr, qr);
          const T df[3] = {out[0] - pr[0], out[1] - pr[1], out[2] - pr[2]};
          qr[1] = -qr[1]; qr[2] = -qr[2]; qr[3] = -qr[3];
          rot_vec_quat(r, df, qr);
          out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
        }
      }
      else if (t == DMC_SENS_SUBTREELINVEL) {}   // written by subtree_linvel_sensors()
      else if (L.d.nrf && t == DMC_SENS_RANGEFINDER) {
        // mj_ray along the site's z axis: nearest visible geom that is not on the site's own body
        const int sb = MI(site_bodyid)[id];
        T v[3], sp[3], q[4], e[3] = {0, 0, 1}, vec[3];
        mul_mat_vec3(v, S(xmat) + 9*sb, MRS(site_pos) + 3*id);
        for (int k = 0; k < 3; k++) sp[k] = S(xpos)[3*sb + k] + v[k];
        mul_quat(q, SG(xquat) + 4*sb, MRS(site_quat) + 4*id);
        rot_vec_quat(vec, e, q);
        T best = -1;
        for (int g = 0; g < L.d.ngeom; g++) {
          if (MI(geom_bodyid)[g] == sb || MI(geom_invisible)[g]) continue;
          const T x = ray_geom_any(S(geom_xpos) + 3*g, SG(geom_xmat) + 9*g, MR(geom_size) + 3*g, sp, vec, MI(geom_type)[g]);
          if (x >= 0 && (best < 0 || x < best)) best = x;
        }
        out[0] = best;
      }
      else if (t == DMC_SENS_FRAMEQUAT) {
        const int otr = MI(sensor_objtype)[i], ot = otr & 255;
        T q[4];
        if (ot == DMC_OBJ_SITE) mul_quat(q, SG(xquat) + 4*MI(site_bodyid)[id], MRS(site_quat) + 4*id);
        else if (ot == DMC_OBJ_GEOM) mul_quat(q, SG(xquat) + 4*MI(geom_bodyid)[id], MRC(geom_quat) + 4*id);
        else if (ot == DMC_OBJ_BODY) mul_quat(q, SG(xquat) + 4*id, MRC(body_iquat) + 4*id);
        else for (int k = 0; k < 4; k++) q[k] = SG(xquat)[4*id + k];
        if (otr >> 16) {      // conj(q_ref) q
          T qr[4], pr[3], qq[4];
          sensor_ref_pose((otr >> 8) & 255, (otr >> 16) - 1, pr, qr);
          qr[1] = -qr[1]; qr[2] = -qr[2]; qr[3] = -qr[3];
          mul_quat(qq, qr, q);
          for (int k = 0; k < 4; k++) q[k] = qq[k];
        }
        for (int k = 0; k < 4; k++) out[k] = q[k];
      }
      else if (t == DMC_SENS_FRAMELINVEL || t == DMC_SENS_FRAMEANGVEL) {
        // mj_objectVelocity in world orientation at the object's frame origin
        const int otr = MI(sensor_objtype)[i], ot = otr & 255;
        const int b = ot == DMC_OBJ_SITE ? MI(site_bodyid)[id] : (ot == DMC_OBJ_GEOM ? MI(geom_bodyid)[id] : id);
        T p[3];
        if (ot == DMC_OBJ_SITE) {
          T v[3]; mul_mat_vec3(v, S(xmat) + 9*b, MRS(site_pos) + 3*id);
          for (int k = 0; k < 3; k++) p[k] = S(xpos)[3*b + k] + v[k];
        } else {
          const T* pp = ot == DMC_OBJ_GEOM ? S(geom_xpos) + 3*id : (ot == DMC_OBJ_BODY ? S(xipos) + 3*id : S(xpos) + 3*id);
          for (int k = 0; k < 3; k++) p[k] = pp[k];
        }
        const T* rc = S(subtree_com) + 3*MI(body_rootid)[b];
        const T dif[3] = {p[0] - rc[0], p[1] - rc[1], p[2] - rc[2]};
        const T* cv = S(cvel) + 6*b;
        T tmp[3], w[3], v[3];
        cross3(tmp, dif, cv);
        for (int k = 0; k < 3; k++) { w[k] = cv[k]; v[k] = cv[3 + k] - tmp[k]; }
        if (otr >> 16) {
          // relative to a moving reference frame (mj_sensorVel): R_ref' (v - v_ref + r x w_ref), R_ref' (w - w_ref)
          const int rt = (otr >> 8) & 255, rid = (otr >> 16) - 1;
          const int rb = rt == DMC_OBJ_SITE ? MI(site_bodyid)[rid] : (rt == DMC_OBJ_GEOM ? MI(geom_bodyid)[rid] : rid);
          T qr[4], pr[3], cr[3], rel[3], res[3];
          sensor_ref_pose(rt, rid, pr, qr);
          const T* rcr = S(subtree_com) + 3*MI(body_rootid)[rb];
          const T difr[3] = {pr[0] - rcr[0], pr[1] - rcr[1], pr[2] - rcr[2]};
          const T* cvr = S(cvel) + 6*rb;
          if (t == DMC_SENS_FRAMEANGVEL) for (int k = 0; k < 3; k++) rel[k] = w[k] - cvr[k];
          else {
            const T r[3] = {p[0] - pr[0], p[1] - pr[1], p[2] - pr[2]};
            cross3(tmp, difr, cvr);
            cross3(cr, r, cvr);
            for (int k = 0; k < 3; k++) rel[k] = v[k] - (cvr[3 + k] - tmp[k]) + cr[k];
          }
          qr[1] = -qr[1]; qr[2] = -qr[2]; qr[3] = -qr[3];
          rot_vec_quat(res, rel, qr);
          for (int k = 0; k < 3; k++) { w[k] = res[k]; v[k] = res[k]; }
        }
        for (int k = 0; k < 3; k++) out[k] = t == DMC_SENS_FRAMEANGVEL ? w[k] : v[k];
      }
      else if (t == DMC_SENS_VELOCIMETER || t == DMC_SENS_GYRO) {
        const int b = MI(site_bodyid)[id]; T v[3], q[4], m[9], sp[3], v6[6];
        mul_mat_vec3(v, S(xmat) + 9*b, MRS(site_pos) + 3*id);
        for (int k = 0; k < 3; k++) sp[k] = S(xpos)[3*b + k] + v[k];
        mul_quat(q, SG(xquat) + 4*b, MRS(site_quat) + 4*id);
        quat2mat(m, q);
        object_velocity(b, sp, m, v6);
        for (int k = 0; k < 3; k++) out[k] = t == DMC_SENS_GYRO ? v6[k] : v6[3 + k];
      }
    }
    DMC_WSYNC();
  }

  // ---- actuation + smooth acceleration ---------------------------------------------
  DMC_DEV void fwd_actuation(bool disable_actuation) {
    const int nv = L.d.nv, nu = L.d.nu;
    if (disable_actuation || (o.disableflags & DMC_DSBL_ACTUATION)) {
      FOR_LANES(i, nv) S(qfrc_actuator)[i] = 0;
      FOR_LANES(i, nu) S(actuator_force)[i] = 0;
      if (L.d.na) FOR_LANES(i, L.d.na) S(act_dot)[i] = 0;
      DMC_WSYNC();
      return;
    }
    int bad = 0;
    FOR_LANES(i, nu) if (t_bad(S(ctrl)[i])) bad = 1;
    bad = group_max<LPE>(bad);
    if (bad) {
      DMC_WSYNC();
      FOR_LANES(i, nu) S(ctrl)[i] = 0;
      if (lane == 0) SI(imisc)[IM_WARN + DMC_WARN_BADCTRL]++;
      DMC_WSYNC();
    }
    FOR_LANES(i, nu) {
      T ctrl = S(ctrl)[i];
      const int fl = MI(act_flags)[i];
      if ((fl & ACTF_CTRLLIMITED) && !(o.disableflags & DMC_DSBL_CLAMPCTRL))
        ctrl = t_max(MRC(act_ctrlrange)[2*i], t_min(MRC(act_ctrlrange)[2*i + 1], ctrl));
      const T* gp = MRC(act_gainprm) + 3*i; const T* bp = MRC(act_biasprm) + 3*i;
      const T gear = MRC(act_gear)[i];
      T len, vel;
      if (L.d.nwrap && (fl & ACTF_TENDON)) {   // fixed tendon: linear combination of joint coordinates
        const int t = MI(act_dof)[i];
        len = 0; vel = 0;
        for (int w = MI(tendon_adr)[t]; w < MI(tendon_adr)[t] + MI(tendon_num)[t]; w++) {
          len += MR(wrap_prm)[w] * S(qpos)[MI(wrap_qpos)[w]]; vel += MR(wrap_prm)[w] * S(qvel)[MI(wrap_dof)[w]];
        }
        len *= gear; vel *= gear;
      } else { len = gear * S(qpos)[MI(act_qpos)[i]]; vel = gear * S(qvel)[MI(act_dof)[i]]; }
      T gain = gp[0], bias = 0;
      if (fl & ACTF_GAIN_AFFINE) gain = gp[0] + gp[1]*len + gp[2]*vel;
      if (fl & ACTF_BIAS_AFFINE) bias = bp[0] + bp[1]*len + bp[2]*vel;
      // actuators with dynamics: the activation drives the gain; integrator: act_dot = ctrl,
      // filter / filterexact: act_dot = (ctrl - act) / tau
      T input = ctrl;
      if (L.d.na) { if (fl & ACTF_DYN_ANY) {
        const int k = MI(act_adr)[i];
        const T act = S(act)[k];
        S(act_dot)[k] = (fl & ACTF_DYN_INTEGRATOR) ? ctrl : (ctrl - act) / t_max((T)DMC_MINVAL, MR(act_dynprm)[i]);
        input = act;
      } }
      T force = gain*input + bias;
      if (fl & ACTF_FORCELIMITED) force = t_max(MRC(act_forcerange)[2*i], t_min(MRC(act_forcerange)[2*i + 1], force));
      S(actuator_force)[i] = force;
    }
    DMC_WSYNC();
    FOR_LANES(dd, nv) {
      T f = 0;
      for (int i = 0; i < nu; i++) {
        if (L.d.nwrap && (MI(act_flags)[i] & ACTF_TENDON)) {
          const int t = MI(act_dof)[i];
          for (int w = MI(tendon_adr)[t]; w < MI(tendon_adr)[t] + MI(tendon_num)[t]; w++)
            if (MI(wrap_dof)[w] == dd) f += MRC(act_gear)[i] * MR(wrap_prm)[w] * S(actuator_force)[i];
        } else if (MI(act_dof)[i] == dd) f += MRC(act_gear)[i] * S(actuator_force)[i];
      }
      S(qfrc_actuator)[dd] = f;
    }
    DMC_WSYNC();
  }
  // xfrc_applied[b] = [force, torque] at the COM of body b, from global memory (group-uniform address per b)
  DMC_DEV void load_xfrc(int b, T* xf) const {
    const T* p = (const T*)o.xfrc + (size_t)6*b*o.xfrc_B + SI(imisc)[IM_ENV];
    for (int k = 0; k < 6; k++) xf[k] = p[(size_t)k*o.xfrc_B];
  }
  DMC_DEV void fwd_acceleration() {
    FOR_LANES(i, L.d.nv) S(qfrc_smooth)[i] = S(qfrc_passive)[i] - S(qfrc_bias)[i] + S(qfrc_applied)[i] + S(qfrc_actuator)[i];
    if (o.xfrc) {
      // mj_xfrcAccumulate: qfrc_smooth += J_b' [f; tau] with the body Jacobian at xipos, through the com-based motion
      // axes: column k of J at point p is (cdof_lin + cdof_ang x (p - com_root), cdof_ang) for dofs above body b
      FOR_LANES(i, L.d.nv) {
        T acc = 0;
        for (int b = 1; b < L.d.nbody; b++) {
          if (!dof_in_chain(MI(body_lastdof)[b], i)) continue;
          T xf[6]; load_xfrc(b, xf);
          const T* rc = S(subtree_com) + 3*MI(body_rootid)[b];
          const T dif[3] = {S(xipos)[3*b] - rc[0], S(xipos)[3*b + 1] - rc[1], S(xipos)[3*b + 2] - rc[2]};
          T t[3]; cross3(t, dif, xf);                      // (p - c) x f
          const T* cd = S(cdof) + 6*i;
          acc += cd[0]*(xf[3] + t[0]) + cd[1]*(xf[4] + t[1]) + cd[2]*(xf[5] + t[2]) + cd[3]*xf[0] + cd[4]*xf[1] + cd[5]*xf[2];
        }
        S(qfrc_smooth)[i] += acc;
      }
    }
    DMC_WSYNC();
    chol_solve(S(qacc_smooth), M_factor(), S(qfrc_smooth), L.d.nv, true);
  }

  // ---- Newton solver on the primal (mj_fwdConstraint / mj_solNewton) -----------------
  // efc_state/efc_force from efc_jar; returns the constraint cost (group-uniform)
  // track != null: also record the active set and report whether it differs from
  // the one H was last factored for (0/1, group-uniform)
  // elliptic cones: a frictional contact is one block of `dim` rows whose cost is
  // the squared distance of the residual to the dual cone -- three zones
  // (top: satisfied, bottom: per-row quadratic, middle: 0.5 Dm (N - mu T)^2).
  // The lane that owns the block's first row handles the whole block.  For the
  // middle zone it also leaves the rank structure of the block Hessian
  //   Hc = Dm [ p p' + c (diag(g) - w w') ]   (p, w: combinations of the block's rows)
  // in efc_ca / efc_cb / efc_cg for newton_gradient.
  // Round 5: every load sits at the top of the trip without a predicate (clamped indices), the row types are told apart on
  // registers, and the block of a frictional contact is six statically indexed rows under a `j < dim` guard -- the
  // run-time trip counts had put U[] / fj[] in scratch memory and every row's (jar, D) behind its own LDS round trip
  // inside three nested exec-mask regions.  Same expressions, same order of the sums.
  DMC_DEV T constraint_update_ell(int nefc, int* track) {
    T cost = 0;
    int changed = 0;
    for (int i = lane; i < nefc; i += LPE) {
      const int tid = SI(efc_tid)[i], ty = EFC_TYPE(tid), id = EFC_ID(tid);
      const T jar = S(efc_jar)[i], D = S(efc_D)[i];
      const int old = SI(efc_active)[i];
      const bool ell = ty == EFC_ELLIPTIC;
      const int c = ell ? id : 0;
      const int r0 = SI(con_efc)[c], info = SI(con_info)[c];
      const bool head = ell && i == r0;
      const int rb = head ? r0 : i;      // (a lane that heads no block re-reads its own row)
      T jr[6], Dr[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {      // (rows past the block are read and ignored; the device reads at most five words past the array, inside the scratch)
#ifdef DMC_HOST_EMU
        const int rj = rb + j < L.d.njmax ? rb + j : rb;
#else
        const int rj = rb + j;
#endif
        jr[j] = S(efc_jar)[rj]; Dr[j] = S(efc_D)[rj];
      }
      const T floss = MR(dof_frictionloss)[ty == EFC_FRICTION ? id : 0];
      const T* fr = MR(prm_friction) + 3*(head ? prm_of_info(info) : 0);
      const T fr0 = fr[0], fr1 = fr[1], fr2 = fr[2];
      if (!ell) {
        // (the new total of every zone is formed as `cost + term`, the expression the branches had: the same contraction, the
        // same bits -- the iteration counts of the 62-dof model move with the last bit of this sum)
        const T fq = -D*jar, tq = cost + (T)0.5*D*jar*jar;
        int st; T force, tot;
        if (ty == EFC_EQUALITY) { st = EFC_ST_QUADRATIC; force = fq; tot = tq; }      // two-sided: always quadratic
        else if (ty == EFC_FRICTION) {
          // Huber cost: quadratic for |jar| < R*floss, linear (force saturated at +-floss) outside
          const T f = floss, rf = f / D;
          const T tn = cost + f*((T)-0.5*rf - jar), tp = cost + f*((T)-0.5*rf + jar);
          if (jar <= -rf) { st = EFC_ST_LINEARNEG; force = f; tot = tn; }
          else if (jar >= rf) { st = EFC_ST_LINEARPOS; force = -f; tot = tp; }
          else { st = EFC_ST_QUADRATIC; force = fq; tot = tq; }
        } else {
          const bool act = jar < 0;
          st = act ? 1 : 0; force = act ? fq : (T)0; tot = act ? tq : cost;
        }
        S(efc_force)[i] = force; cost = tot;
        if (track) { changed |= old != st; SI(efc_active)[i] = st; }
        continue;
      }
      if (!head) continue;
      const int dim = info & 0xff;
      const T D0 = Dr[0];
      const T mu = fr0 * t_sqrt(D0 / Dr[1]);   // regularised cone: mu sqrt(R1/R0)
      T U[6], fj[6], Tn = 0;
      U[0] = jr[0]*mu; fj[0] = mu;
#pragma unroll
      for (int j = 1; j < 6; j++) {
        fj[j] = j < 3 ? fr0 : (j == 3 ? fr1 : fr2);
        U[j] = jr[j]*fj[j];
        if (j < dim) Tn += U[j]*U[j];
      }
      Tn = t_sqrt(Tn);
      const T N = U[0];
      int st;
      if (N >= mu*Tn || (Tn <= 0 && N >= 0)) {
        st = EFC_ST_SATISFIED;
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) S(efc_force)[r0 + j] = 0;
      } else if (mu*N + Tn <= 0 || (Tn <= 0 && N < 0)) {
        st = EFC_ST_QUADRATIC;
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) { S(efc_force)[r0 + j] = -Dr[j]*jr[j]; cost += (T)0.5*Dr[j]*jr[j]*jr[j]; }
      } else {
        st = EFC_ST_CONE;
        const T Dm = D0 / t_max((T)DMC_MINVAL, mu*mu*(1 + mu*mu)), NT = N - mu*Tn;
        cost += (T)0.5*Dm*NT*NT;
        const T f0 = -Dm*NT*mu;
        S(efc_force)[r0] = f0;
        const T cc = -NT*mu/Tn;   // > 0
        S(efc_ca)[r0] = mu; S(efc_cb)[r0] = Dm*cc; S(efc_cg)[r0] = Dm;
#pragma unroll
        for (int j = 1; j < 6; j++) if (j < dim) {
          const T uh = U[j]/Tn;
          S(efc_force)[r0 + j] = -f0/Tn * U[j]*fj[j];
          S(efc_ca)[r0 + j] = -mu*uh*fj[j]; S(efc_cb)[r0 + j] = uh*fj[j]; S(efc_cg)[r0 + j] = Dm*cc*fj[j]*fj[j];
        }
      }
      if (track) {
        // a cone-zone Hessian depends on the residual itself, not just on the zone
        if (old != st || st == EFC_ST_CONE) changed = 1;
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) SI(efc_active)[r0 + j] = st;
      }
    }
    cost = group_sum<LPE>(cost);
    if (track) *track = group_max<LPE>(changed);
    DMC_WSYNC();
    return cost;
  }
  DMC_DEV void hess_assemble(int nefc, const RowMap& rm) {
    const int nv = L.d.nv, K = L.d.kmax;
    if (L.d.jfull && !L.d.msparse) {
      // small models (dense M, every row dense): ONE pass over the packed triangle, entry (i, j) = M(i, j) + the rows in
      // order -- no zero fill, no scatter, no fence before the factorisation's own
      for (int idx = lane; idx < L.d.ntri; idx += LPE) {
        int i, j;
        tri_unrank(idx, nv, &i, &j);
        T h = S(qM)[i*nv + j];
        if (!L.d.elliptic) {
          // four rows per trip: their loads are issued together (one LDS round trip instead of four dependent ones --
          // the row count is a run-time number, so the loop is not unrolled for us), then accumulated in row order
          for (int r = 0; r < nefc; r += 4) {
            int st[4]; T ji[4], jj[4], dd[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int rr = r + u < nefc ? r + u : r;
              st[u] = r + u < nefc ? SI(efc_active)[rr] : 0;
              ji[u] = S(efc_Jd)[rr*nv + i]; jj[u] = S(efc_Jd)[rr*nv + j]; dd[u] = S(efc_D)[rr];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (st[u] == EFC_ST_QUADRATIC && ji[u] != 0) h += (dd[u]*ji[u]) * jj[u];
          }
          S(qLH)[idx] = h;
          continue;
        }
        for (int r = 0; r < nefc; r++) {
          const int st = SI(efc_active)[r];
          if (L.d.elliptic && st == EFC_ST_CONE) {
            const int dim = con_dim(EFC_ID(SI(efc_tid)[r]));
            T Pi = 0, Pj = 0, Wi = 0, Wj = 0, g = 0;
            for (int a = 0; a < dim; a++) {
              const T ji = S(efc_Jd)[(r + a)*nv + i], jj = S(efc_Jd)[(r + a)*nv + j];
              const T ca = S(efc_ca)[r + a];
              Pi += ca*ji; Pj += ca*jj;
              if (a) { const T cb = S(efc_cb)[r + a]; Wi += cb*ji; Wj += cb*jj; g += S(efc_cg)[r + a]*ji*jj; }
            }
            h += S(efc_cg)[r]*(Pi*Pj) - S(efc_cb)[r]*(Wi*Wj) + g;
            r += dim - 1;
            continue;
          }
          const T ji = S(efc_Jd)[r*nv + i], jj = S(efc_Jd)[r*nv + j], dd = S(efc_D)[r];
          if (st == EFC_ST_QUADRATIC && ji != 0) h += (dd*ji) * jj;
        }
        S(qLH)[idx] = h;
      }
      DMC_WSYNC();
      return;
    }
    FOR_LANES(i, nv) {
      T dsum = 0;
      for (int r = rm.s0; r < rm.tl0; r += 4) {      // four rows per trip: their loads are issued together
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool in = r + u < rm.tl0;
          const int rr = in ? r + u : r;      // (a padding row re-reads the trip's first row: no load behind a predicate)
          const int st = SI(efc_active)[rr], tid = SI(efc_tid)[rr];
          const T dd = S(efc_D)[rr], t_ = dsum + dd;
          dsum = (in && st == EFC_ST_QUADRATIC && simple_dof(tid) == i) ? t_ : dsum;
        }
      }
      S(sv_Mgrad)[i] = dsum;
    }
    scatter_M(S(sv_Mgrad), (T)1, S(qLH));      // (fences inside: sv_Mgrad is complete before it is read)
    if (L.d.njdense && (rm.s0 > 0 || rm.c0 > rm.tl0)) {
      for (int idx = lane; idx < L.d.ntri; idx += LPE) {
        int i, j;
        tri_unrank(idx, nv, &i, &j);
        T h = S(qLH)[idx];
        for (int r = 0; r < rm.c0; r++) {
          if (r >= rm.s0 && r < rm.tl0) continue;
          const int st = SI(efc_active)[r];
          if (L.d.jfull && L.d.elliptic && st == EFC_ST_CONE) {      // all-dense models: the cone block of an elliptic contact
            const int dim = con_dim(EFC_ID(SI(efc_tid)[r]));
            T Pi = 0, Pj = 0, Wi = 0, Wj = 0, g = 0;
            for (int a = 0; a < dim; a++) {
              const T ji = S(efc_Jd)[(r + a)*nv + i], jj = S(efc_Jd)[(r + a)*nv + j];
              const T ca = S(efc_ca)[r + a];
              Pi += ca*ji; Pj += ca*jj;
              if (a) { const T cb = S(efc_cb)[r + a]; Wi += cb*ji; Wj += cb*jj; g += S(efc_cg)[r + a]*ji*jj; }
            }
            h += S(efc_cg)[r]*(Pi*Pj) - S(efc_cb)[r]*(Wi*Wj) + g;
            r += dim - 1;
            continue;
          }
          if (st != EFC_ST_QUADRATIC) continue;
          const T* jr = dense_row(r, rm);
          const T ji = jr[i];
          if (ji != 0) h += (S(efc_D)[r]*ji) * jr[j];
        }
        S(qLH)[idx] = h;
      }
      DMC_WSYNC();
    }
    const int ncon = rm.c0 < nefc ? SI(imisc)[IM_NCON] : 0;
    const auto Jc_base = Jc();
    for (int c = 0; c < ncon; c++) {
      const int r0 = SI(con_efc)[c];
      if (r0 < 0) continue;
      const int nrow = contact_rows(con_dim(c));
      const int st0 = SI(efc_active)[r0];
      const bool cone = L.d.elliptic && st0 == EFC_ST_CONE;
      // the row loops run to the compile-time bound maxrow (model-specialised kernels) under a predicate, so
      // that the loads of all rows are in flight together
      bool any = cone;
      for (int q = 0; q < L.d.maxrow; q++) if (q < nrow) any = any || SI(efc_active)[r0 + q] == EFC_ST_QUADRATIC;
      if (!any) continue;               // group-uniform: a contact whose rows are all inactive adds nothing
      const int kc = con_ndof(c), npair = (kc*(kc + 1)) >> 1;
      const unsigned char* dofs = con_dof_list(c);
      const auto J = Jc_base + (r0 - rm.c0)*K;
      for (int t = lane; t < npair; t += LPE) {
        // slot pair (a >= b) of the block, row-major
        int a = (int)((sqrtf(8.0f*(float)t + 1.0f) - 1.0f)*0.5f);
        if (((a + 1)*(a + 2)) >> 1 <= t) a++;
        else if ((a*(a + 1)) >> 1 > t) a--;
        const int b2 = t - ((a*(a + 1)) >> 1);
        const int i = dofs[a], j = dofs[b2];
        T acc = 0;
        if (cone) {
          T Pi = 0, Pj = 0, Wi = 0, Wj = 0, g = 0;
          for (int q = 0; q < L.d.maxrow; q++) if (q < nrow) {
            const T ji = J[q*K + a], jj = J[q*K + b2];
            const T ca = S(efc_ca)[r0 + q];
            Pi += ca*ji; Pj += ca*jj;
            if (q) { const T cb = S(efc_cb)[r0 + q]; Wi += cb*ji; Wj += cb*jj; g += S(efc_cg)[r0 + q]*ji*jj; }
          }
          acc = S(efc_cg)[r0]*(Pi*Pj) - S(efc_cb)[r0]*(Wi*Wj) + g;
        } else {
          for (int q = 0; q < L.d.maxrow; q++) if (q < nrow && SI(efc_active)[r0 + q] == EFC_ST_QUADRATIC) {
            const T ji = J[q*K + a];
            const T jj = J[q*K + b2], t_ = acc + (S(efc_D)[r0 + q]*ji) * jj;
            acc = ji != 0 ? t_ : acc;
          }
        }
        S(qLH)[tri_at(i, j, nv)] += acc;
      }
      DMC_WSYNC();
    }
  }
  // H of a split solve (h_split): only the entries inside the trees' diagonal blocks exist, and the side-by-side
  // factorisation and substitutions read nothing else.  One lane per in-tree entry (StepDims::ntreetri: 105 for five 6-dof
  // trees, against 465 packed entries zero-filled, the scatter of M and one FENCED pass per contact -- 1.8 k instructions
  // per assembly on the soccer model, where an instruction of a lone wave costs ~9 cycles): M(i, j) from the sparse M, the
  // diagonal sum of the one-nonzero rows, then the contacts IN ORDER -- a contact adds to the entry when both dofs are in
  // its mask -- so every entry is the same sum in the same order as hess_assemble's.
  DMC_DEV void hess_assemble_split(int nefc, const RowMap& rm) {
    const int nv = L.d.nv, K = L.d.kmax;
    FOR_LANES(i, nv) {
      T dsum = 0;
      for (int r = rm.s0; r < rm.tl0; r += 4) {      // four rows per trip: their loads are issued together
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool in = r + u < rm.tl0;
          const int rr = in ? r + u : r;      // (a padding row re-reads the trip's first row: no load behind a predicate)
          const int st = SI(efc_active)[rr], tid = SI(efc_tid)[rr];
          const T dd = S(efc_D)[rr], t_ = dsum + dd;
          dsum = (in && st == EFC_ST_QUADRATIC && simple_dof(tid) == i) ? t_ : dsum;
        }
      }
      S(sv_Mgrad)[i] = dsum;
    }
#ifdef DMC_HOST_EMU
    FOR_LANES(t, L.d.ntri) S(qLH)[t] = 0;      // (the emulation factors the whole matrix)
#endif
    DMC_WSYNC();
    const int ncon = rm.c0 < nefc ? SI(imisc)[IM_NCON] : 0;
    const auto Jc_base = Jc();
    // two entries per lane and trip (t and t + LPE): the walk over the contacts -- a chain of dependent look-ups per
    // contact (row, state, mask, Jacobian entries) -- is paid once for both
    for (int t = lane; t < L.d.ntreetri; t += 2*LPE) {
      const bool two = t + LPE < L.d.ntreetri;
      const int pk0 = MI(tree_tri)[t], pk1 = MI(tree_tri)[two ? t + LPE : t];
      const int mix0 = MI(tree_trim)[t], mix1 = MI(tree_trim)[two ? t + LPE : t];
      const int i0 = pk0 & 0xffff, j0 = pk0 >> 16, i1 = pk1 & 0xffff, j1 = pk1 >> 16;
      T h0 = mix0 >= 0 ? qMs()[mix0] : (T)0, h1 = mix1 >= 0 ? qMs()[mix1] : (T)0;
      if (i0 == j0) h0 += S(sv_Mgrad)[i0];
      if (i1 == j1) h1 += S(sv_Mgrad)[i1];
      for (int c = 0; c < ncon; c++) {
        const int r0 = SI(con_efc)[c];
        if (r0 < 0) continue;
        const int nrow = contact_rows(con_dim(c));
        const bool cone = L.d.elliptic && SI(efc_active)[r0] == EFC_ST_CONE;
        bool any = cone;
        for (int q = 0; q < L.d.maxrow; q++) if (q < nrow) any = any || SI(efc_active)[r0 + q] == EFC_ST_QUADRATIC;
        if (!any) continue;               // group-uniform
        const unsigned lo = con_mask_lo(c), hi = con_mask_hi(c);
        const int a0 = mask_slot(lo, hi, i0), b0 = mask_slot(lo, hi, j0), a1 = mask_slot(lo, hi, i1), b1 = mask_slot(lo, hi, j1);
        const bool in0 = a0 >= 0 && b0 >= 0, in1 = two && a1 >= 0 && b1 >= 0;
        // (no lane-dependent skip: an entry outside the contact's mask reads slot 0 and drops the sum -- the skip put the
        // global loads of the contact's Jacobian behind an exec-mask sequence)
        const int sa0 = in0 ? a0 : 0, sb0 = in0 ? b0 : 0, sa1 = in1 ? a1 : 0, sb1 = in1 ? b1 : 0;
        const auto J = Jc_base + (r0 - rm.c0)*K;
        T acc0 = 0, acc1 = 0;
        if (cone) {
          T Pi0 = 0, Pj0 = 0, Wi0 = 0, Wj0 = 0, g0 = 0, Pi1 = 0, Pj1 = 0, Wi1 = 0, Wj1 = 0, g1 = 0;
          for (int q = 0; q < L.d.maxrow; q++) if (q < nrow) {
            const T ji0 = J[q*K + sa0], jj0 = J[q*K + sb0], ji1 = J[q*K + sa1], jj1 = J[q*K + sb1];
            const T ca = S(efc_ca)[r0 + q];
            Pi0 += ca*ji0; Pj0 += ca*jj0; Pi1 += ca*ji1; Pj1 += ca*jj1;
            if (q) {
              const T cb = S(efc_cb)[r0 + q], cg = S(efc_cg)[r0 + q];
              Wi0 += cb*ji0; Wj0 += cb*jj0; g0 += cg*ji0*jj0;
              Wi1 += cb*ji1; Wj1 += cb*jj1; g1 += cg*ji1*jj1;
            }
          }
          const T cg0 = S(efc_cg)[r0], cb0 = S(efc_cb)[r0];
          acc0 = cg0*(Pi0*Pj0) - cb0*(Wi0*Wj0) + g0;
          acc1 = cg0*(Pi1*Pj1) - cb0*(Wi1*Wj1) + g1;
        } else {
          for (int q = 0; q < L.d.maxrow; q++) if (q < nrow && SI(efc_active)[r0 + q] == EFC_ST_QUADRATIC) {
            const T dq = S(efc_D)[r0 + q];
            const T ji0 = J[q*K + sa0], ji1 = J[q*K + sa1];
            const T jj0 = J[q*K + sb0], jj1 = J[q*K + sb1], t0 = acc0 + (dq*ji0) * jj0, t1 = acc1 + (dq*ji1) * jj1;
            acc0 = ji0 != 0 ? t0 : acc0; acc1 = ji1 != 0 ? t1 : acc1;
          }
        }
        if (in0) h0 += acc0;
        if (in1) h1 += acc1;
      }
      S(qLH)[tri_at(i0, j0, nv)] = h0;
      if (two) S(qLH)[tri_at(i1, j1, nv)] = h1;
    }
    DMC_WSYNC();
  }
  // Once per line search (elliptic models): what a frictional contact contributes to every evaluation depends on alpha
  // only through N = U0 + alpha V0 and T^2 = UU + alpha (2 UV + alpha VV); those aggregates, the regularised mu and the
  // three sums of the bottom (fully quadratic) zone are parked in the contact's first three rows of the cone-coefficient
  // arrays, which are dead between the Hessian that read them and the next constraint_update that rewrites them.  An
  // evaluation then is one round of loads per contact instead of a chain of index look-ups, a square root, a division
  // and 3 dim loads (the line search was 20 % of the soccer step and 12 % of the 62-dof one).
  DMC_DEV void ls_prepare_ell(int nefc) {
    for (int i = lane; i < nefc; i += LPE) {
      const int tid = SI(efc_tid)[i];
      if (EFC_TYPE(tid) != EFC_ELLIPTIC) continue;
      const int c = EFC_ID(tid), r0 = SI(con_efc)[c];
      if (i != r0) continue;
      const int dim = con_dim(c);
      const T* fr = MR(prm_friction) + 3*con_prm(c);
      const T D0 = S(efc_D)[r0];
      const T mu = fr[0] * t_sqrt(D0 / S(efc_D)[r0 + 1]);
      const T U0 = S(efc_jar)[r0]*mu, V0 = S(efc_jv)[r0]*mu;
      T UU = 0, UV = 0, VV = 0, b0 = 0, b1 = 0, b2 = 0;
      for (int j = 0; j < dim; j++) {
        const T jar = S(efc_jar)[r0 + j], jv = S(efc_jv)[r0 + j], D = S(efc_D)[r0 + j], dj0 = D*jar;
        b0 += (T)0.5*jar*dj0; b1 += jv*dj0; b2 += (T)0.5*D*jv*jv;
        if (j) {
          const T f = fr[j < 3 ? 0 : (j == 3 ? 1 : 2)];
          const T u = jar*f, v = jv*f;
          UU += u*u; UV += u*v; VV += v*v;
        }
      }
      S(efc_ca)[r0] = U0; S(efc_ca)[r0 + 1] = V0; S(efc_ca)[r0 + 2] = UU;
      S(efc_cb)[r0] = UV; S(efc_cb)[r0 + 1] = VV; S(efc_cb)[r0 + 2] = mu;
      S(efc_cg)[r0] = b0; S(efc_cg)[r0 + 1] = b1; S(efc_cg)[r0 + 2] = b2;
    }
    DMC_WSYNC();
  }
  DMC_DEV bool anchored() const { return ls_anchored<T>() && (!general_rows() || L.d.nv > 32); }
  // Middle-zone cost of a frictional contact, anchored form (ls_anchored): the cost at alpha minus the cost at 0 minus
  // alpha x its slope at 0, and the slope at alpha minus the slope at 0.  With NT = N - mu T the cost is 1/2 Dm NT^2; late
  // in a solve NT moves by less than an fp32 ulp of itself along the whole search, so NT(alpha) - NT(0) must not be
  // formed from the two values: T(alpha) - T(0) = alpha (2 UV + alpha VV) / (T(alpha) + T(0)) =: alpha W, and with
  // E = VV T0 - UV W:   dNT = alpha (V0 - mu W),   NT1(alpha) - NT1(0) = -mu alpha E / (T T0),
  // dNT - alpha NT1(0) = -mu alpha^2 E / ((T + T0) T0)   -- products only.
  DMC_DEV static void cone_middle_anchored(T a, bool middle, bool middle0, T Dm, T NT, T NT1, T Tn, T NT0, T T0,
                                           T V0, T UV, T VV, T mu, T* cc, T* cd0) {
    if (middle && middle0) {
      const T W = (2*UV + a*VV)/(Tn + T0), E = VV*T0 - UV*W;
      const T dNT = a*(V0 - mu*W), dNT1 = -mu*a*E/(Tn*T0), dlin = -mu*a*a*E/((Tn + T0)*T0);
      *cc += Dm*(NT0*dlin + (T)0.5*dNT*dNT);
      *cd0 += Dm*(NT0*dNT1 + dNT*NT1);
    } else {
      if (middle) { *cc += (T)0.5*Dm*NT*NT; *cd0 += Dm*NT*NT1; }
      if (middle0) { const T l0 = Dm*NT0*(V0 - mu*UV/T0); *cc -= (T)0.5*Dm*NT0*NT0 + a*l0; *cd0 -= l0; }      // l0: the slope at alpha = 0
    }
  }
  // line-search point for elliptic models: quadratic rows as in ls_eval_lds, plus
  // the non-quadratic middle-zone term of every frictional contact
  DMC_DEV void ls_eval_ell(dmc::LSPoint<T>* p, const T* qg, int nefc) {
    const T a = p->alpha;
    constexpr bool rel = ls_relative<T>();      // cost relative to alpha = 0 (fp32), see ls_relative
    const bool anch = anchored();               // ... with the linear term anchored on grad . search, see ls_anchored
    T q0 = 0, q1 = 0, q2 = 0, cc = 0, cd0 = 0, cd1 = 0;
    for (int i = lane; i < nefc; i += LPE) {
      const int tid = SI(efc_tid)[i];
      if (EFC_TYPE(tid) == EFC_EQUALITY) {
        const T jar = S(efc_jar)[i], jv = S(efc_jv)[i], D = S(efc_D)[i], dj0 = D*jar;
        if (!rel) q0 += (T)0.5*jar*dj0;
        if (!anch) q1 += jv*dj0;
        q2 += (T)0.5*D*jv*jv;
        continue;
      }
      if (EFC_TYPE(tid) == EFC_FRICTION) {
        const T jar = S(efc_jar)[i], jv = S(efc_jv)[i], D = S(efc_D)[i];
        const T f = MR(dof_frictionloss)[EFC_ID(tid)], rf = f / D, x = jar + a*jv;
        const int za = x <= -rf ? -1 : (x >= rf ? 1 : 0), z0 = jar <= -rf ? -1 : (jar >= rf ? 1 : 0);
        if (za < 0) { q0 += f*((T)-0.5*rf - jar); if (!anch) q1 += -f*jv; }
        else if (za > 0) { q0 += f*((T)-0.5*rf + jar); if (!anch) q1 += f*jv; }
        else { const T dj0 = D*jar; q0 += (T)0.5*jar*dj0; if (!anch) q1 += jv*dj0; q2 += (T)0.5*D*jv*jv; }
        if (rel) {      // minus the row's cost at alpha = 0 (Huber: linear outside |jar| < rf)
          if (z0 < 0) q0 -= f*((T)-0.5*rf - jar);
          else if (z0 > 0) q0 -= f*((T)-0.5*rf + jar);
          else q0 -= (T)0.5*jar*D*jar;
        }
        if (anch && za != z0) q1 += (za < 0 ? -f*jv : (za > 0 ? f*jv : jv*(D*jar))) - (z0 < 0 ? -f*jv : (z0 > 0 ? f*jv : jv*(D*jar)));
        continue;
      }
      if (EFC_TYPE(tid) != EFC_ELLIPTIC) {
        const T jar = S(efc_jar)[i], jv = S(efc_jv)[i];
        if (rel) {
          const bool act_a = jar + a*jv < 0, act_0 = jar < 0;
          if (act_a | act_0) {
            const T D = S(efc_D)[i], dj0 = D*jar;
            if (act_a) { if (!anch) q1 += jv*dj0; q2 += (T)0.5*D*jv*jv; }
            if (act_a != act_0) { q0 += (act_a ? (T)0.5 : (T)-0.5)*jar*dj0; if (anch) q1 += act_a ? jv*dj0 : -(jv*dj0); }
          }
        } else if (jar + a*jv < 0) { const T D = S(efc_D)[i], dj0 = D*jar; q0 += (T)0.5*jar*dj0; q1 += jv*dj0; q2 += (T)0.5*D*jv*jv; }
        continue;
      }
      // first row of a frictional contact: its alpha-independent aggregates were parked by ls_prepare_ell
      const int r0 = i;
      if (SI(con_efc)[EFC_ID(tid)] != r0) continue;
      const T U0 = S(efc_ca)[r0], V0 = S(efc_ca)[r0 + 1], UU = S(efc_ca)[r0 + 2];
      const T UV = S(efc_cb)[r0], VV = S(efc_cb)[r0 + 1], mu = S(efc_cb)[r0 + 2];
      const T N = U0 + a*V0, Tsqr = UU + a*(2*UV + a*VV);
      bool bottom = false, middle = false;
      T NT = 0, Dm = 0, NT1 = 0, Tn = 0;
      if (Tsqr <= 0) bottom = N < 0;
      else {
        Tn = t_sqrt(Tsqr);
        if (N >= mu*Tn) {}
        else if (mu*N + Tn <= 0) bottom = true;
        else {
          middle = true;
          Dm = S(efc_D)[r0] / t_max((T)DMC_MINVAL, mu*mu*(1 + mu*mu));
          const T N1 = V0, T1 = (UV + a*VV)/Tn, T2 = VV/Tn - (UV + a*VV)*T1/(Tn*Tn);
          NT1 = N1 - mu*T1;
          NT = N - mu*Tn;
          if (!rel) cc += (T)0.5*Dm*NT*NT;
          if (!anch) cd0 += Dm*NT*NT1;
          cd1 += Dm*(NT1*NT1 - NT*mu*T2);
        }
      }
      if (!rel) { if (bottom) { q0 += S(efc_cg)[r0]; q1 += S(efc_cg)[r0 + 1]; q2 += S(efc_cg)[r0 + 2]; } continue; }
      // relative form: the contact's cost at alpha minus its cost at 0, by the pair of zones
      bool bottom0 = false, middle0 = false;
      T NT0 = 0, T0 = 0;
      if (UU <= 0) bottom0 = U0 < 0;
      else {
        T0 = t_sqrt(UU);
        if (U0 >= mu*T0) {}
        else if (mu*U0 + T0 <= 0) bottom0 = true;
        else { middle0 = true; NT0 = U0 - mu*T0; }
      }
      if (bottom) { if (!anch || !bottom0) q1 += S(efc_cg)[r0 + 1]; q2 += S(efc_cg)[r0 + 2]; if (!bottom0) q0 += S(efc_cg)[r0]; }
      else if (bottom0) q0 -= S(efc_cg)[r0];
      if (anch && bottom0 && !bottom) q1 -= S(efc_cg)[r0 + 1];
      if (middle | middle0) {
        if (!middle0 || !middle) Dm = S(efc_D)[r0] / t_max((T)DMC_MINVAL, mu*mu*(1 + mu*mu));
        if (!anch) cc += (T)0.5*Dm*(NT - NT0)*(NT + NT0);      // 1/2 Dm (NT^2 - NT0^2); a zone that is not the middle one has NT = 0
        else cone_middle_anchored(a, middle, middle0, Dm, NT, NT1, Tn, NT0, T0, V0, UV, VV, mu, &cc, &cd0);
      }
    }
#ifdef DMC_HOST_EMU
    if (getenv("DMC_EMU_TRACE_LS2")) fprintf(stderr, "      eval_ell a %.9e: q0 %.6e q1rows %.6e q2rows %.6e cc %.6e cd0 %.6e | qg %.6e %.6e %.6e\n", (double)a, (double)q0, (double)q1, (double)q2, (double)cc, (double)cd0, (double)qg[0], (double)qg[1], (double)qg[2]);
#endif
    q0 = group_sum<LPE>(q0) + (rel ? (T)0 : qg[0]); q1 = group_sum<LPE>(q1) + qg[1]; q2 = group_sum<LPE>(q2) + qg[2];
    cc = group_sum<LPE>(cc); cd0 = group_sum<LPE>(cd0); cd1 = group_sum<LPE>(cd1);
    p->cost = a*a*q2 + a*q1 + q0 + cc;
    p->d0 = 2*a*q2 + q1 + cd0;
    p->d1 = 2*q2 + cd1;
    if (p->d1 <= 0) p->d1 = (T)DMC_MINVAL;
  }
  DMC_DEV T constraint_update(int nefc, int* track = nullptr) {
    if (general_rows()) return constraint_update_ell(nefc, track);
    T cost = 0;
    int changed = 0;
    for (int i = lane; i < nefc; i += LPE) {
      // (branch-free: D is read beside jar, not after the test on it -- a predicated read is a second LDS round trip
      // behind an exec-mask sequence; the sums see the same terms, a zero where the row is satisfied)
      const T jar = S(efc_jar)[i], D = S(efc_D)[i];
      const int old = track ? SI(efc_active)[i] : 0;
      const int act = jar < 0;
      const T dj = D*jar;
      S(efc_force)[i] = act ? -dj : (T)0;
      if (sizeof(T) == 4) cost += act ? (T)0.5*dj*jar : (T)0; else if (act) cost += (T)0.5*D*jar*jar;
      if (track) { changed |= old != act; SI(efc_active)[i] = act; }
    }
    cost = group_sum<LPE>(cost);
    if (track) *track = group_max<LPE>(changed);
    DMC_WSYNC();
    return cost;
  }
  DMC_DEV T gauss_cost() {
    T g = 0;
    FOR_LANES(i, L.d.nv) g += (S(sv_Ma)[i] - S(qfrc_smooth)[i]) * (S(qacc)[i] - S(qacc_smooth)[i]);
    return (T)0.5 * group_sum<LPE>(g);
  }
  // res = M v.  Small models keep the dense M: a row product.  Tree-sparse models (nv > 16) never touch M: M v is the
  // joint-space force that produces acceleration v at zero velocity without gravity, so it is evaluated the way
  // mj_rne would (composite-inertia identity  M_ij = cdof_i . (sum_{d in subtree(i) & subtree(j)} cinert_d) cdof_j):
  //   A. every body b in parallel: spatial acceleration a_b = sum of cdof_j v_j over the dofs j on the path to b
  //      (the ancestor bit mask of its last dof), then its inertial force f_b = cinert_b a_b;
  //   B. leaves to root, one tree level per fence: f_b += f_children (descending child order, as crb_mass_matrix);
  //   C. every dof in parallel: res_i = cdof_i . f_body(i) + armature_i v_i.
  // Cost: nlevel fences and O(depth) work per lane, instead of nv serial look-ups through the ancestor masks per row
  // (root rows of a humanoid are dense): on the 62-dof model 6.6 products per physics step were 11 % of the step.
  DMC_DEV void mul_M(T* res, const T* v) {
    const int nv = L.d.nv;
    if (!L.d.msparse) { FOR_LANES(i, nv) res[i] = dot_n(S(qM) + i*nv, v, nv); return; }
    const int nb = L.d.nbody;
    T* bf = S(sv_bf);
    FOR_LANES(b, nb) {
      T a[6] = {0, 0, 0, 0, 0, 0}, f[6] = {0, 0, 0, 0, 0, 0};
      const int ld = b > 0 ? MI(body_lastdof)[b] : -1;
      if (ld >= 0) {
        unsigned lo = (unsigned)MI(dof_anc_lo)[ld], hi = nv > 32 ? (unsigned)MI(dof_anc_hi)[ld] : 0u;
        while (lo) { const int j = __builtin_ctz(lo); lo &= lo - 1; const T vj = v[j]; const T* cd = S(cdof) + 6*j; for (int k = 0; k < 6; k++) a[k] += cd[k]*vj; }
        while (hi) { const int j = 32 + __builtin_ctz(hi); hi &= hi - 1; const T vj = v[j]; const T* cd = S(cdof) + 6*j; for (int k = 0; k < 6; k++) a[k] += cd[k]*vj; }
        mul_inert_vec(f, SG(cinert) + 10*b, a);
      }
      for (int k = 0; k < 6; k++) bf[6*b + k] = f[k];
    }
    DMC_WSYNC();
    if (L.d.dfs) {      // B + C in one pass: the dof's lane sums the inertial forces over the subtree of its body (an id range)
      FOR_LANES(i, nv) {
        const int b = MI(dof_bodyid)[i], e = MI(body_subend)[b];
        T f[6] = {0, 0, 0, 0, 0, 0};
        for (int c = b; c < e; c++) for (int a = 0; a < 6; a++) f[a] += bf[6*c + a];
        res[i] = dot_n(S(cdof) + 6*i, f, 6) + MR(dof_armature)[i]*v[i];
      }
      return;
    }
    for (int lev = L.d.nlevel - 2; lev >= 0; lev--) {
      const int a0 = MI(level_adr)[lev], cnt = MI(level_adr)[lev + 1] - a0;
      for (int idx = lane; idx < cnt*6; idx += LPE) {
        const int b = MI(level_body)[a0 + idx/6], comp = idx % 6;
        const int c0 = MI(child_adr)[b], c1 = MI(child_adr)[b + 1];
        if (c1 > c0) {
          T acc = bf[6*b + comp];
          for (int c = c0; c < c1; c++) acc += bf[6*MI(child_list)[c] + comp];
          bf[6*b + comp] = acc;
        }
      }
      DMC_WSYNC();
    }
    FOR_LANES(i, nv) res[i] = dot_n(S(cdof) + 6*i, bf + 6*MI(dof_bodyid)[i], 6) + MR(dof_armature)[i]*v[i];
  }
  DMC_DEV void jar_from(const T* qacc, int nefc) {
    const RowMap rm = row_map();
    for (int i = lane; i < nefc; i += LPE) S(efc_jar)[i] = row_dot(i, qacc, rm) - S(efc_aref)[i];
  }
  // qfrc_constraint = J' efc_force, rows visited in order
  DMC_DEV void constraint_force_to_joint(int nefc) {
    const int nv = L.d.nv, K = L.d.kmax;
    const RowMap rm = row_map();
    if (L.d.jfull) {      // every row dense: one loop, rows in order
      FOR_LANES(i, nv) {
        T f = 0;
        for (int r = 0; r < nefc; r += 4) {      // four rows per trip, loads issued together, summed in row order
          T fr[4], jr[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const int rr = r + u < nefc ? r + u : r; const T w = S(efc_force)[rr]; fr[u] = r + u < nefc ? w : (T)0; jr[u] = S(efc_Jd)[rr*nv + i]; }
#pragma unroll
          for (int u = 0; u < 4; u++) { if (sizeof(T) == 4) f += jr[u]*fr[u]; else f = fr[u] != 0 ? f + jr[u]*fr[u] : f; }      // (a zero force adds a zero: no branch)
        }
        S(qfrc_constraint)[i] = f;
      }
      return;
    }
    const int ncon = rm.c0 < nefc ? SI(imisc)[IM_NCON] : 0;
    const auto Jc_base = Jc();
    FOR_LANES(i, nv) {
      T f = 0;
      if (L.d.njdense) for (int r = 0; r < rm.s0; r++) { const T fr = S(efc_force)[r]; if (fr != 0) f += S(efc_Jd)[r*nv + i]*fr; }
      // (no lane-dependent branch around a load: a dof outside the contact's mask reads slot 0 and keeps its sum -- what sat
      // behind `if (slot < 0) continue` was an exec-mask sequence and, for the models whose contact rows live in global
      // memory, one dependent global round trip per row inside it)
      for (int r = rm.s0; r < rm.tl0; r++) {
        const T fr = S(efc_force)[r];
        const int tid = SI(efc_tid)[r];
        const T t_ = f + simple_sign(tid)*fr;
        f = (fr != 0 && simple_dof(tid) == i) ? t_ : f;
      }
      if (L.d.njdense) for (int r = rm.tl0; r < rm.c0; r++) { const T fr = S(efc_force)[r], t_ = f + S(efc_Jd)[(rm.s0 + r - rm.tl0)*nv + i]*fr; f = fr != 0 ? t_ : f; }
      for (int c = 0; c < ncon; c++) {
        const int r0 = SI(con_efc)[c];
        if (r0 < 0) continue;      // (group-uniform)
        const int slot = mask_slot(con_mask_lo(c), con_mask_hi(c), i);
        const int nrow = contact_rows(con_dim(c));
        const auto jc = Jc_base + (r0 - rm.c0)*K + (slot < 0 ? 0 : slot);
        for (int q = 0; q < L.d.maxrow; q++) if (q < nrow) { const T fr = S(efc_force)[r0 + q], t_ = f + jc[q*K]*fr; f = (slot >= 0 && fr != 0) ? t_ : f; }
      }
      S(qfrc_constraint)[i] = f;
    }
  }
  // H = M + J' D_active J assembled and factored IN REGISTERS (model-specialised small dense models: dense M, every row
  // dense, no elliptic cones): lane i builds row i of H -- the rows of J are read as broadcasts, four constraint rows
  // per trip, accumulated in row order exactly like hess_assemble's entries -- and the factorisation is
  // chol_factor_rows' on the same registers: no packed-index arithmetic, no trip of H through LDS, no fence between
  // assembly and factorisation (they were 3 us of a 5 us newton_gradient on the cheetah).
#ifndef DMC_HOST_EMU
  template <int N>
  DMC_DEV void hess_factor_rows(int nefc) {
    static_assert(N >= 1 && N <= LPE, "one lane per matrix row");
    const bool own = lane < N;
    const int i = own ? lane : 0;
    T a[N];
    // Branch-free (round 5): every lane builds its WHOLE row (M is stored symmetric; the entries right of the diagonal are
    // never read by the elimination's valid lanes), the row's loads carry no predicate -- a lane outside the matrix reads
    // row 0, a trip's padding rows re-read the trip's first row with a zero weight -- and a row that is not in the
    // quadratic zone enters with weight zero: a + 0 * J_rj is a, to the bit (the Jacobian is finite).  What the compiler made
    // of the predicated form was a v_cmp / exec-mask / load / restore sequence per entry and three DEPENDENT LDS round
    // trips per constraint row (state, then J_ri, then D) -- 2.7 us per assembly + factorisation of the 9-dof model, of which
    // the arithmetic is a tenth.  nv <= 16: J_rj comes from lane j's own J_rj by a DPP row broadcast instead of nine
    // wave-uniform LDS reads per row.
#pragma unroll
    for (int j = 0; j < N; j++) a[j] = S(qM)[i*N + j];
    for (int r = 0; r < nefc; r += 4) {
      T c[4], ji[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int rr = r + u < nefc ? r + u : r;
        const int st = SI(efc_active)[rr];
        ji[u] = S(efc_Jd)[rr*N + i];
        const T w = (r + u < nefc && st == EFC_ST_QUADRATIC) ? S(efc_D)[rr] : (T)0;
        c[u] = w*ji[u];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
#pragma unroll
        for (int j = 0; j < N; j++) {
          T jj;
          if constexpr (N <= 16 && LPE >= 16) jj = bcast_rows<LPE, N>(ji[u], j); else jj = S(efc_Jd)[(r + u < nefc ? r + u : r)*N + j];
          a[j] += c[u] * jj;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
      T akk = bcast_rows<LPE, N>(a[k], k);
      if (akk < (T)DMC_MINVAL) akk = (T)DMC_MINVAL;
      const T inv = t_rsqrt(akk);
      const T lik = a[k] * inv;
#pragma unroll
      for (int j = k + 1; j < N; j++) { const T ljk = bcast_rows<LPE, N>(lik, j); a[j] = nmsub<(N <= 16)>(a[j], lik, ljk); }
      a[k] = lane == k ? inv : lik;
    }
    store_factor_rows<T, N>((DMC_LDS T*)S(qLH), a, lane);
    DMC_WSYNC();
  }
#endif
  // refactor == 0: the active set is unchanged, so H and its factor (still in qLH)
  // are reused -- identical values, none of the O(nv^3) work
  DMC_DEV void newton_gradient(int nefc, int refactor) {
    const int nv = L.d.nv;
    constraint_force_to_joint(nefc);
    DMC_WSYNC();
    FOR_LANES(i, nv) S(sv_grad)[i] = S(sv_Ma)[i] - S(qfrc_smooth)[i] - S(qfrc_constraint)[i];
    DMC_PROF(PROF_SOL_GRAD);
    // CG (mj_solPrimal with flg_Newton = 0): the gradient preconditioned with M^-1 -- the factor mj_factorM left
    // (no Hessian is ever assembled, so it is still in place)
    if (L.d.cg) { DMC_WSYNC(); chol_solve(S(sv_Mgrad), M_factor(), S(sv_grad), nv, true); DMC_PROF(PROF_SOLVE); return; }
    if (!refactor) { DMC_WSYNC(); chol_solve(S(sv_Mgrad), S(qLH), S(sv_grad), nv, hsplit); DMC_PROF(PROF_SOLVE); return; }
#ifndef DMC_HOST_EMU
    if constexpr (LS::kNV > 0 && LS::kNV <= 16 && LS::kNV <= LPE) {
      if (L.d.jfull && !L.d.msparse && !L.d.elliptic) {
        DMC_WSYNC();
        hess_factor_rows<LS::kNV>(nefc);
        DMC_PROF(PROF_FACTOR);
        chol_solve(S(sv_Mgrad), S(qLH), S(sv_grad), nv);
        DMC_PROF(PROF_SOLVE);
        return;
      }
    }
#endif
#ifdef DMC_HOST_EMU
    // The emulation factors H in full either way.  With hsplit set it checks what the device relies on: the full
    // assembly has an exact zero wherever an entry joins two trees, and the in-tree assembly produces the same bits.
    hess_assemble(nefc, row_map());
    if (hsplit) {
      T* full = (T*)malloc(sizeof(T)*L.d.ntri);
      for (int t = 0; t < L.d.ntri; t++) full[t] = S(qLH)[t];
      for (int i = 0; i < nv; i++) for (int j = 0; j < MI(dof_tree0)[i]; j++) if (full[tri_at(i, j, nv)] != 0) {
        fprintf(stderr, "h_split: H(%d, %d) = %g joins two trees\n", i, j, (double)full[tri_at(i, j, nv)]); abort();
      }
      hess_assemble_split(nefc, row_map());
      for (int t = 0; t < L.d.ntri; t++) if (memcmp(&full[t], &S(qLH)[t], sizeof(T))) {
        fprintf(stderr, "hess_assemble_split: packed entry %d is %.17g, hess_assemble has %.17g\n", t, (double)S(qLH)[t], (double)full[t]); abort();
      }
      free(full);
    }
#else
#ifndef DMC_NO_TREE_SPLIT
    if (hsplit) hess_assemble_split(nefc, row_map()); else
#endif
    hess_assemble(nefc, row_map());
#endif
    DMC_PROF(PROF_HESS);
    chol_factor_inplace(S(qLH), nv, hsplit);
    DMC_PROF(PROF_FACTOR);
    chol_solve(S(sv_Mgrad), S(qLH), S(sv_grad), nv, hsplit);
    DMC_PROF(PROF_SOLVE);
  }
  // Is H = M + J'DJ of this solve block diagonal over the kinematic trees?  Yes unless a constraint row moves two of them:
  // a contact between bodies of two trees (its dof mask leaves the tree of its first dof), or any dense row (equalities,
  // tendon limits -- not looked at, taken as coupling).  Dof-friction and joint-limit rows have one nonzero.  Decided once
  // per solve from the rows that exist, active or not.
  DMC_DEV bool h_split(int nefc) {
    if (!L.d.treemax) return false;
    const RowMap rm = row_map();
    if (rm.s0 > 0 || rm.c0 > rm.tl0) return false;
    const int ncon = rm.c0 < nefc ? SI(imisc)[IM_NCON] : 0;
    int coupled = 0;
    for (int c = lane; c < ncon; c += LPE) {
      if (SI(con_efc)[c] < 0) continue;
      const unsigned lo = con_mask_lo(c), hi = con_mask_hi(c);
      if (!(lo | hi)) continue;
      const int first = lo ? __builtin_ctz(lo) : 32 + __builtin_ctz(hi);
      const int last = hi ? 63 - __builtin_clz(hi) : 31 - __builtin_clz(lo);
      if (last >= MI(dof_tree1)[first]) coupled = 1;
    }
    return group_max<LPE>(coupled) == 0;
  }
  typedef dmc::LSPoint<T> LSPoint;
  // Line search on REGISTER-RESIDENT rows: with at most one constraint row per lane (nefc <= LPE) and only one-sided
  // quadratic rows, the lane keeps its row's (jar, jv, D) for all evaluations of a search -- an evaluation is a few
  // VALU ops and three reductions, with no call and no LDS round trip (a Newton iteration of the cheetah spends 3.5-6 us
  // in its 2-8 dependent evaluations; the launch waits for the wave with the most iterations).  Same arithmetic as
  // ls_eval_lds for the lane's one row.
  // The same for the general row types (equalities, dof friction loss, elliptic contacts): `kind` says what the lane's row
  // is; the first row of a frictional contact carries the contact's alpha-independent aggregates (ls_prepare_ell's,
  // computed into registers instead of being parked in LDS), its zone at alpha = 0 and the middle-zone scale; the
  // contact's other rows carry nothing.  An evaluation of the soccer / 62-dof models was a chain of LDS look-ups (row
  // type -> contact -> first row -> nine aggregates) per row before the arithmetic started: 22 % / 14 % of their step.
  enum { LSK_NONE = 0, LSK_EQUALITY, LSK_FRICTION, LSK_ONESIDED, LSK_CONE };
  struct LSRows { T jar, jv, D; bool on; bool gen; int kind; T f, rf, U0, V0, UU, UV, VV, mu, b0, b1, b2, Dm, NT0, T0; bool bottom0, middle0; };
  // (round 5: loads first and unpredicated, the contact block as six statically indexed rows -- see constraint_update_ell)
  DMC_DEV void ls_load_gen(LSRows& g, int nefc) {
    g.gen = true; g.kind = LSK_NONE;
    g.f = g.rf = g.U0 = g.V0 = g.UU = g.UV = g.VV = g.mu = g.b0 = g.b1 = g.b2 = g.Dm = g.NT0 = g.T0 = 0; g.bottom0 = g.middle0 = false;
    const bool in = lane < nefc;
    const int i = in ? lane : 0;
    const int tid = SI(efc_tid)[i], ty = EFC_TYPE(tid), id = EFC_ID(tid);
    const bool ell = in && ty == EFC_ELLIPTIC;
    const int c = ell ? id : 0;
    const int r0 = SI(con_efc)[c], info = SI(con_info)[c];
    const bool head = ell && i == r0;
    T jr[6], jvv[6], Dr[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
#ifdef DMC_HOST_EMU
      const int rj = i + j < L.d.njmax ? i + j : i;
#else
      const int rj = i + j;
#endif
      jr[j] = S(efc_jar)[rj]; jvv[j] = S(efc_jv)[rj]; Dr[j] = S(efc_D)[rj];
    }
    const T floss = MR(dof_frictionloss)[(in && ty == EFC_FRICTION) ? id : 0];
    const T* fr = MR(prm_friction) + 3*(head ? prm_of_info(info) : 0);
    const T fr0 = fr[0], fr1 = fr[1], fr2 = fr[2];
    if (!in) return;
    g.jar = jr[0]; g.jv = jvv[0]; g.D = Dr[0];
    if (ty == EFC_EQUALITY) { g.kind = LSK_EQUALITY; return; }
    if (ty == EFC_FRICTION) { g.kind = LSK_FRICTION; g.f = floss; g.rf = g.f / g.D; return; }
    if (ty != EFC_ELLIPTIC) { g.kind = LSK_ONESIDED; return; }
    if (!head) return;
    g.kind = LSK_CONE;
    const int dim = info & 0xff;
    const T D0 = g.D;
    const T mu = fr0 * t_sqrt(D0 / Dr[1]);
    T UU = 0, UV = 0, VV = 0, b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) {
      const T jar = jr[j], jv = jvv[j], D = Dr[j], dj0 = D*jar;
      b0 += (T)0.5*jar*dj0; b1 += jv*dj0; b2 += (T)0.5*D*jv*jv;
      if (j) {
        const T f = j < 3 ? fr0 : (j == 3 ? fr1 : fr2);
        const T u = jar*f, v = jv*f;
        UU += u*u; UV += u*v; VV += v*v;
      }
    }
    g.U0 = g.jar*mu; g.V0 = g.jv*mu; g.UU = UU; g.UV = UV; g.VV = VV; g.mu = mu; g.b0 = b0; g.b1 = b1; g.b2 = b2;
    g.Dm = D0 / t_max((T)DMC_MINVAL, mu*mu*(1 + mu*mu));
    if (UU <= 0) g.bottom0 = g.U0 < 0;      // the contact's zone at alpha = 0 (relative form)
    else {
      const T T0 = t_sqrt(UU);
      if (g.U0 >= mu*T0) {}
      else if (mu*g.U0 + T0 <= 0) g.bottom0 = true;
      else { g.middle0 = true; g.NT0 = g.U0 - mu*T0; g.T0 = T0; }      // (T0: the anchored form's reference point)
    }
  }
  // ls_eval_ell's arithmetic for the lane's one row
  DMC_DEV void ls_eval_gen(LSPoint* p, const T* qg, const LSRows& g) {
    const T a = p->alpha;
    constexpr bool rel = ls_relative<T>();
    const bool anch = anchored();
    T q0 = 0, q1 = 0, q2 = 0, cc = 0, cd0 = 0, cd1 = 0;
    if (g.kind == LSK_EQUALITY) {
      const T dj0 = g.D*g.jar;
      if (!rel) q0 += (T)0.5*g.jar*dj0;
      if (!anch) q1 += g.jv*dj0;
      q2 += (T)0.5*g.D*g.jv*g.jv;
    } else if (g.kind == LSK_FRICTION) {
      const T jar = g.jar, jv = g.jv, D = g.D, f = g.f, rf = g.rf, x = jar + a*jv;
      const int za = x <= -rf ? -1 : (x >= rf ? 1 : 0), z0 = jar <= -rf ? -1 : (jar >= rf ? 1 : 0);
      if (za < 0) { q0 += f*((T)-0.5*rf - jar); if (!anch) q1 += -f*jv; }
      else if (za > 0) { q0 += f*((T)-0.5*rf + jar); if (!anch) q1 += f*jv; }
      else { const T dj0 = D*jar; q0 += (T)0.5*jar*dj0; if (!anch) q1 += jv*dj0; q2 += (T)0.5*D*jv*jv; }
      if (rel) {
        if (z0 < 0) q0 -= f*((T)-0.5*rf - jar);
        else if (z0 > 0) q0 -= f*((T)-0.5*rf + jar);
        else q0 -= (T)0.5*jar*D*jar;
      }
      if (anch && za != z0) q1 += (za < 0 ? -f*jv : (za > 0 ? f*jv : jv*(D*jar))) - (z0 < 0 ? -f*jv : (z0 > 0 ? f*jv : jv*(D*jar)));
    } else if (g.kind == LSK_ONESIDED) {
      const T jar = g.jar, jv = g.jv;
      if (rel) {
        const bool act_a = jar + a*jv < 0, act_0 = jar < 0;
        if (act_a | act_0) {
          const T D = g.D, dj0 = D*jar;
          if (act_a) { if (!anch) q1 += jv*dj0; q2 += (T)0.5*D*jv*jv; }
          if (act_a != act_0) { q0 += (act_a ? (T)0.5 : (T)-0.5)*jar*dj0; if (anch) q1 += act_a ? jv*dj0 : -(jv*dj0); }
        }
      } else if (jar + a*jv < 0) { const T D = g.D, dj0 = D*jar; q0 += (T)0.5*jar*dj0; q1 += jv*dj0; q2 += (T)0.5*D*jv*jv; }
    } else if (g.kind == LSK_CONE) {
      const T U0 = g.U0, V0 = g.V0, UU = g.UU, UV = g.UV, VV = g.VV, mu = g.mu, Dm = g.Dm;
      const T N = U0 + a*V0, Tsqr = UU + a*(2*UV + a*VV);
      bool bottom = false, middle = false;
      T NT = 0, NT1 = 0, Tn = 0;
      if (Tsqr <= 0) bottom = N < 0;
      else {
        Tn = t_sqrt(Tsqr);
        if (N >= mu*Tn) {}
        else if (mu*N + Tn <= 0) bottom = true;
        else {
          middle = true;
          const T N1 = V0, T1 = (UV + a*VV)/Tn, T2 = VV/Tn - (UV + a*VV)*T1/(Tn*Tn);
          NT1 = N1 - mu*T1;
          NT = N - mu*Tn;
          if (!rel) cc += (T)0.5*Dm*NT*NT;
          if (!anch) cd0 += Dm*NT*NT1;
          cd1 += Dm*(NT1*NT1 - NT*mu*T2);
        }
      }
      if (!rel) { if (bottom) { q0 += g.b0; q1 += g.b1; q2 += g.b2; } }
      else {
        if (bottom) { if (!anch || !g.bottom0) q1 += g.b1; q2 += g.b2; if (!g.bottom0) q0 += g.b0; }
        else if (g.bottom0) q0 -= g.b0;
        if (anch && g.bottom0 && !bottom) q1 -= g.b1;
        if (middle | g.middle0) {
          if (!anch) cc += (T)0.5*Dm*(NT - g.NT0)*(NT + g.NT0);
          else cone_middle_anchored(a, middle, g.middle0, Dm, NT, NT1, Tn, g.NT0, g.T0, V0, UV, VV, mu, &cc, &cd0);
        }
      }
    }
    q0 = group_sum<LPE>(q0) + (rel ? (T)0 : qg[0]); q1 = group_sum<LPE>(q1) + qg[1]; q2 = group_sum<LPE>(q2) + qg[2];
    cc = group_sum<LPE>(cc); cd0 = group_sum<LPE>(cd0); cd1 = group_sum<LPE>(cd1);
    p->cost = a*a*q2 + a*q1 + q0 + cc;
    p->d0 = 2*a*q2 + q1 + cd0;
    p->d1 = 2*q2 + cd1;
    if (p->d1 <= 0) p->d1 = (T)DMC_MINVAL;
  }
  DMC_DEV void ls_eval(LSPoint* p, const T* qg, int nefc, int* evals, const LSRows& rw) {
#ifdef DMC_HOST_EMU
    emu_ls_counts()[1]++;
#endif
    if (general_rows()) { if (rw.gen) ls_eval_gen(p, qg, rw); else ls_eval_ell(p, qg, nefc); (*evals)++; return; }
    if (rw.on) {
      const T a = p->alpha, jar = rw.jar, jv = rw.jv;
      T q0 = 0, q1 = 0, q2 = 0;
      if (ls_relative<T>()) {
        const bool act_a = jar + a*jv < 0, act_0 = jar < 0;
        if (act_a | act_0) {
          const T D = rw.D, dj0 = D*jar;
          if (act_a) { if (!ls_anchored<T>()) q1 += jv*dj0; q2 += (T)0.5*D*jv*jv; }
          if (act_a != act_0) { q0 += (act_a ? (T)0.5 : (T)-0.5)*jar*dj0; if (ls_anchored<T>()) q1 += act_a ? jv*dj0 : -(jv*dj0); }
        }
      } else if (jar + a*jv < 0) {
        const T D = rw.D, dj0 = D*jar;
        q0 += (T)0.5*jar*dj0; q1 += jv*dj0; q2 += (T)0.5*D*jv*jv;
      }
      q0 = group_sum<LPE>(q0) + (ls_relative<T>() ? (T)0 : qg[0]); q1 = group_sum<LPE>(q1) + qg[1]; q2 = group_sum<LPE>(q2) + qg[2];
      p->cost = a*a*q2 + a*q1 + q0;
      p->d0 = 2*a*q2 + q1;
      p->d1 = 2*q2;
      if (p->d1 <= 0) p->d1 = (T)DMC_MINVAL;
      (*evals)++;
      return;
    }
    const DMC_LSVEC(T) rv = ls_eval_lds<T, LPE>(p->alpha, (const DMC_LDS T*)S(efc_jar), (const DMC_LDS T*)S(efc_jv),
                                                (const DMC_LDS T*)S(efc_D), ls_relative<T>() ? (T)0 : qg[0], qg[1], qg[2], nefc, lane);
    p->alpha = rv[0]; p->cost = rv[1]; p->d0 = rv[2]; p->d1 = rv[3];
    (*evals)++;
  }
  DMC_DEV int ls_update_bracket(LSPoint* p, const LSPoint* cand, LSPoint* pnext, const T* qg, int nefc, int* evals, const LSRows& rw) {
    int flag = 0;
    for (int i = 0; i < 3; i++) {
      if (p->d0 < 0 && cand[i].d0 < 0 && p->d0 < cand[i].d0) { *p = cand[i]; flag = 1; }
      else if (p->d0 > 0 && cand[i].d0 > 0 && p->d0 > cand[i].d0) { *p = cand[i]; flag = 2; }
    }
    if (flag) { pnext->alpha = p->alpha - p->d0/p->d1; ls_eval(pnext, qg, nefc, evals, rw); }
    return flag;
  }
  DMC_DEV T primal_search(int nefc, T gauss, T scale, T* lscost) {
    *lscost = 0;      // cost of the returned point (relative to alpha = 0 in fp32: minus the iteration's improvement)
#ifdef DMC_HOST_EMU
    emu_ls_counts()[0]++;
#endif
    const int nv = L.d.nv;
    mul_M(S(sv_Mv), S(sv_search));
    { const RowMap rm = row_map(); for (int i = lane; i < nefc; i += LPE) S(efc_jv)[i] = row_dot(i, S(sv_search), rm); }
    DMC_WSYNC();
    LSRows rw;
    rw.jar = rw.jv = rw.D = 0; rw.on = false; rw.gen = false; rw.kind = LSK_NONE;
    if (general_rows() && nefc <= LPE && LPE > 1) ls_load_gen(rw, nefc);
    if (L.d.elliptic && !rw.gen) ls_prepare_ell(nefc);
    T a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    const bool anch = anchored();
    FOR_LANES(i, nv) {
      const T sr = S(sv_search)[i];
      if (anch) a1 += sr*S(sv_grad)[i];      // grad . search: the slope at alpha = 0 itself (see ls_anchored)
      else { a1 += sr*S(sv_Ma)[i]; a2 += S(qfrc_smooth)[i]*sr; }
      a3 += sr*S(sv_Mv)[i]; a4 += sr*sr;
    }
    a1 = group_sum<LPE>(a1); if (!anch) a2 = group_sum<LPE>(a2); a3 = group_sum<LPE>(a3); a4 = group_sum<LPE>(a4);
    T qg[3] = {gauss, a1 - a2, (T)0.5*a3};
    const T snorm = t_sqrt(a4);
    if (snorm < (T)DMC_MINVAL) return 0;
    T gtol = o.tolerance * o.ls_tolerance * snorm / scale;
    const int lsmax = o.ls_iterations;
    int evals = 0;
#ifndef DMC_HOST_EMU
    if (!general_rows() && nefc <= LPE) {
      rw.on = true;
      if (lane < nefc) { rw.jar = S(efc_jar)[lane]; rw.jv = S(efc_jv)[lane]; rw.D = S(efc_D)[lane]; }
    }
#endif
    DMC_PROF(PROF_LS_SETUP);
    LSPoint p0, p1, p2, pmid, p1next, p2next;
    p0.alpha = 0; ls_eval(&p0, qg, nefc, &evals, rw);
    // fp32: the slope cannot be resolved below a few ulp of the slope at alpha = 0 (the sums that form it are that
    // large), while MuJoCo's gtol = tolerance * ls_tolerance * |search| / scale sits ~1e-10 below it: fp64 gets there in
    // 4.5 evaluations per search (quadratic convergence), fp32 never did and refined the bracket until no candidate
    // improved it -- 11 evaluations per search on the elliptic models, a quarter of their step.  A point whose slope is
    // within DMC_LS_SLOPE_ULPS ulp of the starting slope is the minimum as far as fp32 can tell: alpha is then off by
    // that relative amount, the cost by its square.
#ifdef DMC_HOST_EMU
    { const char* u = getenv("DMC_LS_SLOPE_ULPS");      // (probe knob of the emulation: scripts/ls_floor_probe.py)
      if (sizeof(T) == 4) gtol = t_max(gtol, (T)((u ? atof(u) : (double)DMC_LS_SLOPE_ULPS) * 1.1920929e-7) * t_abs(p0.d0)); }
#else
    if (sizeof(T) == 4) gtol = t_max(gtol, (T)(DMC_LS_SLOPE_ULPS * 1.1920929e-7) * t_abs(p0.d0));
#endif
    p1.alpha = p0.alpha - p0.d0/p0.d1; ls_eval(&p1, qg, nefc, &evals, rw);
#ifdef DMC_HOST_EMU
    if (getenv("DMC_EMU_TRACE_LS")) fprintf(stderr, "    ls: p0 cost %.6e d0 %.6e d1 %.6e | p1 alpha %.9e cost %.6e d0 %.6e d1 %.6e | gtol %.3e qg1 %.6e\n",
                                           (double)p0.cost, (double)p0.d0, (double)p0.d1, (double)p1.alpha, (double)p1.cost, (double)p1.d0, (double)p1.d1, (double)gtol, (double)qg[1]);
#endif
    if (p0.cost < p1.cost) p1 = p0;
    if (t_abs(p1.d0) < gtol) { *lscost = p1.cost; return p1.alpha; }
    const int dir = p1.d0 < 0 ? 1 : -1;
    int p2update = 0;
    p2 = p1;
    while (p1.d0*dir <= -gtol && evals < lsmax) {
      p2 = p1; p2update = 1;
      p1.alpha -= p1.d0/p1.d1; ls_eval(&p1, qg, nefc, &evals, rw);
      if (t_abs(p1.d0) < gtol) { *lscost = p1.cost; return p1.alpha; }
    }
    if (evals >= lsmax) { *lscost = p1.cost; return p1.alpha; }
    if (!p2update) { *lscost = p1.cost; return p1.alpha; }
    p2next = p1;
    p1next.alpha = p1.alpha - p1.d0/p1.d1; ls_eval(&p1next, qg, nefc, &evals, rw);
    while (evals < lsmax) {
      pmid.alpha = (T)0.5*(p1.alpha + p2.alpha); ls_eval(&pmid, qg, nefc, &evals, rw);
      LSPoint cand[3] = {p1next, p2next, pmid};
      bool found = false; T best_cost = 0, best_alpha = 0;   // first candidate of lowest cost (no dynamic indexing)
      for (int i = 0; i < 3; i++) if (t_abs(cand[i].d0) < gtol && (!found || cand[i].cost < best_cost)) {
        found = true; best_cost = cand[i].cost; best_alpha = cand[i].alpha;
      }
      if (found) { *lscost = best_cost; return best_alpha; }
      const int b1 = ls_update_bracket(&p1, cand, &p1next, qg, nefc, &evals, rw);
      const int b2 = ls_update_bracket(&p2, cand, &p2next, qg, nefc, &evals, rw);
      if (!b1 && !b2) { if (pmid.cost < p0.cost) { *lscost = pmid.cost; return pmid.alpha; } return 0; }
    }
    if (p1.cost <= p2.cost && p1.cost < p0.cost) { *lscost = p1.cost; return p1.alpha; }
    if (p2.cost <= p1.cost && p2.cost < p0.cost) { *lscost = p2.cost; return p2.alpha; }
    return 0;
  }
  // ---- noslip post-solver (mj_solNoSlip; same algorithm and operation order as the oracle's
  // "noslip" section).  Gauss-Seidel in force space over the friction dimensions with R removed:
  // A = J_F M^-1 J_F^T is built once (one substitution per friction row), the residual vector
  // res = J qacc - aref is kept up to date lane-parallel, the per-block scalar maths runs
  // redundantly on every lane (uniform values, no divergence).
  // N is a compile-time constant: every array below is indexed statically and lives in registers (with a
  // run-time size the 20 Newton iterations ran out of scratch memory: 2.5 M cycles per step on the soccer model)
  template <int N>
  DMC_DEV static int qcqp(T* res, const T* Ain, const T* bin, const T* dd, T r) {
    // Every lane of the group runs this scalar Newton iteration redundantly (Gauss-Seidel is sequential), so its
    // instruction count IS the noslip time (31 % of the soccer step).  fp32: one hardware reciprocal square root per
    // pivot serves both L[i][i] and every division by it (the exact form costs ~10 correctly rounded divisions /
    // roots per trip), and the stopping tests are floored at what fp32 resolves (|v|^2 - r^2 cannot get within
    // 1e-10 of zero when r^2 ~ 1e4); fp64 keeps MuJoCo's sequence operation for operation.
#ifdef DMC_EXACT_QCQP
    constexpr bool fast = false;
#else
    constexpr bool fast = sizeof(T) == 4;
#endif
    T A[N*N], b[N], Lc[N*N], inv[N], v[N], pv[N], la = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { v[i] = 0; pv[i] = 0; inv[i] = 0; b[i] = bin[i]*dd[i]; }
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
      for (int j = 0; j < N; j++) { A[i*N + j] = Ain[i*N + j]*dd[i]*dd[j]; Lc[i*N + j] = 0; }
    const T val_floor = fast ? t_max((T)1e-10, (T)(16*1.1920929e-7)*r*r) : (T)1e-10;
    bool fail = false;
    for (int iter = 0; iter < 20; iter++) {
#pragma unroll
      for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
          T t = A[i*N + j] + (i == j ? la : (T)0);
#pragma unroll
          for (int k = 0; k < j; k++) t -= Lc[i*N + k]*Lc[j*N + k];
          if (i == j) {
            if (t < (T)1e-10) fail = true;
            const T tc = t_max(t, (T)1e-30);
            if (fast) { inv[i] = t_rsqrt(tc); Lc[i*N + i] = tc*inv[i]; }
            else Lc[i*N + i] = t_sqrt(tc);
          } else Lc[i*N + j] = fast ? t*inv[j] : t/Lc[j*N + j];
        }
      if (fail) break;
#pragma unroll
      for (int i = 0; i < N; i++) {
        T t = -b[i];
#pragma unroll
        for (int k = 0; k < i; k++) t -= Lc[i*N + k]*v[k];
        v[i] = fast ? t*inv[i] : t/Lc[i*N + i];
      }
#pragma unroll
      for (int i = N - 1; i >= 0; i--) {
        T t = v[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) t -= Lc[k*N + i]*v[k];
        v[i] = fast ? t*inv[i] : t/Lc[i*N + i];
      }
      T val = -r*r;
#pragma unroll
      for (int i = 0; i < N; i++) val += v[i]*v[i];
      if (val < val_floor) break;
#pragma unroll
      for (int i = 0; i < N; i++) {
        T t = v[i];
#pragma unroll
        for (int k = 0; k < i; k++) t -= Lc[i*N + k]*pv[k];
        pv[i] = fast ? t*inv[i] : t/Lc[i*N + i];
      }
      T deriv = 0;
#pragma unroll
      for (int i = 0; i < N; i++) deriv += pv[i]*pv[i];
      deriv *= -2;
      const T delta = -val/deriv;
      if (delta < (fast ? t_max((T)1e-10, (T)(4*1.1920929e-7)*la) : (T)1e-10)) break;
      la += delta;
#ifdef DMC_HOST_EMU
      emu_ls_counts()[5]++;
#endif
    }
#ifdef DMC_HOST_EMU
    emu_ls_counts()[4]++;
#endif
    if (fail) {
#pragma unroll
      for (int i = 0; i < N; i++) res[i] = 0;
      return 0;
    }
#pragma unroll
    for (int i = 0; i < N; i++) res[i] = v[i]*dd[i];
    return la != 0;
  }
  // one Gauss-Seidel block of N friction dimensions starting at slot a; returns the cost change (<= 0)
  // new forces of one block from its A (N x N), old forces and residual: frictionloss dof, pyramidal pair, or the QCQP
  // of an elliptic contact's friction dimensions
  template <int N>
  DMC_DEV void noslip_block_solve(const T* Ac, const T* old, const T* bres, T* fnew, int t, int id) {
    if (N == 1) {
      const T fl = MR(dof_frictionloss)[id];
      fnew[0] = old[0] - bres[0]/Ac[0];
      if (fnew[0] < -fl) fnew[0] = -fl; else if (fnew[0] > fl) fnew[0] = fl;
    } else if (t == EFC_PYRAMIDAL) {
      const T bc0 = bres[0] - Ac[0]*old[0] - Ac[1]*old[N > 1 ? 1 : 0], bc1 = bres[N > 1 ? 1 : 0] - Ac[N > 1 ? N : 0]*old[0] - Ac[N > 1 ? N + 1 : 0]*old[N > 1 ? 1 : 0];
      const T mid = (T)0.5*(old[0] + old[N > 1 ? 1 : 0]);
      const T K1 = Ac[0] + Ac[N > 1 ? N + 1 : 0] - Ac[N > 1 ? 1 : 0] - Ac[N > 1 ? N : 0], K0 = mid*(Ac[0] - Ac[N > 1 ? N + 1 : 0]) + bc0 - bc1;
      T f0, f1;
      if (K1 < (T)DMC_MINVAL) f0 = f1 = mid;
      else {
        const T y = -K0/K1;
        if (y < -mid) { f0 = 0; f1 = 2*mid; }
        else if (y > mid) { f0 = 2*mid; f1 = 0; }
        else { f0 = mid + y; f1 = mid - y; }
      }
      fnew[0] = f0; fnew[N > 1 ? 1 : 0] = f1;
    } else {
      const T* fr3 = MR(prm_friction) + 3*con_prm(id);
      const T fri5[5] = {fr3[0], fr3[0], fr3[1], fr3[2], fr3[2]};
      T fri[N], bc[N];
#pragma unroll
      for (int p = 0; p < N; p++) fri[p] = fri5[p < 5 ? p : 4];
      const T fn = S(efc_force)[SI(con_efc)[id]];
#pragma unroll
      for (int p = 0; p < N; p++) {
        bc[p] = bres[p];
#pragma unroll
        for (int q = 0; q < N; q++) bc[p] -= Ac[p*N + q]*old[q];
      }
      if (fn >= (T)DMC_MINVAL) {
        const int active = qcqp<N>(fnew, Ac, bc, fri, fn);
        if (active) {
          T ss = 0;
#pragma unroll
          for (int p = 0; p < N; p++) ss += (fnew[p]/fri[p])*(fnew[p]/fri[p]);
          ss = t_sqrt(fn*fn / t_max((T)DMC_MINVAL, ss));
#pragma unroll
          for (int p = 0; p < N; p++) fnew[p] *= ss;
        }
      }
    }
  }
  template <int N>
  DMC_DEV T noslip_block(int a, int nf, int t, int id) {
    T Ac[N*N], old[N], bres[N], fnew[N];
#pragma unroll
    for (int p = 0; p < N; p++) {
      old[p] = S(efc_force)[SI(ns_row)[a + p]]; bres[p] = S(ns_res)[a + p]; fnew[p] = 0;
#pragma unroll
      for (int q = 0; q < N; q++) Ac[p*N + q] = ns_A()[(a + p)*L.d.nslip + a + q];
    }
    noslip_block_solve<N>(Ac, old, bres, fnew, t, id);
    // cost change of the block; an update that increases the cost is undone
    T change = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
      T tq = 0;
#pragma unroll
      for (int q = 0; q < N; q++) tq += Ac[p*N + q]*(fnew[q] - old[q]);
      change += (T)0.5*(fnew[p] - old[p])*tq + (fnew[p] - old[p])*bres[p];
    }
    if (change > (T)1e-10) {
#pragma unroll
      for (int p = 0; p < N; p++) fnew[p] = old[p];
      change = 0;
    }
    DMC_WSYNC();
#pragma unroll
    for (int p = 0; p < N; p++) {
      const T delta = fnew[p] - old[p];
      if (lane == 0) S(efc_force)[SI(ns_row)[a + p]] = fnew[p];
      if (delta != 0) { const T* Ap = ns_A() + (a + p)*L.d.nslip; for (int b = lane; b < nf; b += LPE) S(ns_res)[b] += Ap[b]*delta; }
    }
    DMC_WSYNC();
    return change;
  }
  // ---- noslip blocks solved side by side -----------------------------------------------------------------------
  // Gauss-Seidel over the blocks is sequential only where blocks are COUPLED: A = J_F M^-1 J_F' couples two blocks iff
  // their contacts share a kinematic tree (M^-1 is block diagonal over the trees), and for uncoupled ones the cross
  // terms are exact zeros -- the order in which they are solved does not change a bit of the result.  The blocks are
  // put into levels (a block comes after every lower-indexed block it is coupled to); the blocks of one level are
  // solved by one lane each instead of one after the other on every lane (the scalar QCQP iteration IS the noslip
  // time: 31 % of the soccer step, whose four walkers and ball make five independent blocks).  The solving lane
  // updates the residual of its own rows from its registers; only blocks with a partner also sweep their rows of A
  // from global memory over the other rows, in block order, which is the order the sequential sweep adds them in.
  // Up to 16 blocks and trees rooted at bodies 1 .. 63; anything larger keeps the sequential sweep.
  enum { NS_MAXBLK = 16 };
  DMC_DEV void ns_tree_bits(int body, unsigned* lo, unsigned* hi) const {
    if (MI(body_lastdof)[body] < 0) return;      // no dof moves it: nothing of M^-1 behind its Jacobian
    const int r = MI(body_rootid)[body];
    if (r < 32) *lo |= 1u << r; else *hi |= 1u << (r - 32);
  }
  // plans the sweep: returns the number of blocks (0: keep the sequential sweep), *nlev the number of levels
  DMC_DEV int noslip_plan(int nf, int* nlev) {
    *nlev = 0;
    if (L.d.nbody > 64 || L.d.ntree < 2) return 0;      // (one tree: every block is coupled to every other)
    DMC_WSYNC();
    int nb = 0, levmax = 0;
    for (int a = 0; a < nf; ) {
      const int i = SI(ns_row)[a], tid = SI(efc_tid)[i], t = EFC_TYPE(tid), id = EFC_ID(tid);
      int n = 1;
      if (t == EFC_PYRAMIDAL) n = 2;
      else if (t == EFC_ELLIPTIC) n = con_dim(id) - 1;
      if (nb == NS_MAXBLK || a > 255) return 0;
      unsigned lo = 0, hi = 0;
      if (t == EFC_FRICTION) ns_tree_bits(MI(dof_bodyid)[id], &lo, &hi);
      else { ns_tree_bits(con_b1(id), &lo, &hi); ns_tree_bits(con_b2(id), &lo, &hi); }
      int lev = 1, coupled = 0;
      for (int k = 0; k < nb; k++) {
        if (((unsigned)SI(ns_blk)[16 + k] & lo) | ((unsigned)SI(ns_blk)[32 + k] & hi)) {
          const int d = SI(ns_blk)[k], lk = (d >> 16) & 0xff;
          if (lk + 1 > lev) lev = lk + 1;
          coupled = 1;
          if (lane == 0 && !(d >> 24)) SI(ns_blk)[k] = d | (1 << 24);
        }
      }
      if (lane == 0) { SI(ns_blk)[nb] = a | (n << 8) | (lev << 16) | (coupled << 24); SI(ns_blk)[16 + nb] = (int)lo; SI(ns_blk)[32 + nb] = (int)hi; }
      DMC_WSYNC();
      if (lev > levmax) levmax = lev;
      nb++; a += n;
    }
    *nlev = levmax;
    if (levmax == nb) return 0;      // a chain: nothing to solve side by side
    return nb;
  }
  // one block, solved by ONE lane (the others of the group are on other blocks of the same level): noslip_block's
  // arithmetic; the lane writes the forces, updates the residual of the block's own rows and leaves the force changes
  // (efc_jv, dead after the solver) and the cost change (efc_jar[k]) for the group
  template <int N>
  DMC_DEV void noslip_block_lane(int a, int t, int id, int k) {
    T Ac[N*N], old[N], bres[N], fnew[N];
    const T* Ag = ns_A();
#pragma unroll
    for (int p = 0; p < N; p++) {
      old[p] = S(efc_force)[SI(ns_row)[a + p]]; bres[p] = S(ns_res)[a + p]; fnew[p] = 0;
#pragma unroll
      for (int q = 0; q < N; q++) Ac[p*N + q] = Ag[(a + p)*L.d.nslip + a + q];
    }
    noslip_block_solve<N>(Ac, old, bres, fnew, t, id);
    T change = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
      T tq = 0;
#pragma unroll
      for (int q = 0; q < N; q++) tq += Ac[p*N + q]*(fnew[q] - old[q]);
      change += (T)0.5*(fnew[p] - old[p])*tq + (fnew[p] - old[p])*bres[p];
    }
    if (change > (T)1e-10) {
#pragma unroll
      for (int p = 0; p < N; p++) fnew[p] = old[p];
      change = 0;
    }
#pragma unroll
    for (int p = 0; p < N; p++) {
      const T delta = fnew[p] - old[p];
      S(efc_force)[SI(ns_row)[a + p]] = fnew[p];
      S(efc_jv)[a + p] = delta;
      if (delta != 0) {
#pragma unroll
        for (int q = 0; q < N; q++) bres[q] += Ac[p*N + q]*delta;      // rows a .. a + N - 1 of the residual, in the order of the sweep
      }
    }
#pragma unroll
    for (int q = 0; q < N; q++) S(ns_res)[a + q] = bres[q];
    S(efc_jar)[k] = change;
  }
  // sweep of all blocks by levels; returns the sum of the blocks' cost changes, accumulated in block order
  // fp32: a block WITHOUT a partner is at the exact optimum of its own QCQP after the first sweep -- no other block's
  // forces enter its rows of the residual -- and a later sweep would only re-derive the same forces from a residual that
  // differs by the rounding of its own update (1e-7 relative: the noise floor of the fp32 path); such blocks are solved
  // once.  fp64 re-solves them every sweep, operation for operation as the oracle does (noslip_tolerance = 0 models run
  // all their sweeps).
  DMC_DEV T noslip_sweep_levels(int nf, int nb, int nlev, bool first) {
#if defined(DMC_EXACT_QCQP) || defined(DMC_NS_RESWEEP_ALL)
    constexpr bool once = false;
#else
    constexpr bool once = sizeof(T) == 4;
#endif
    for (int lev = 1; lev <= nlev; lev++) {
      FOR_LANES(k, nb) {
        const int d = SI(ns_blk)[k];
        if (((d >> 16) & 0xff) != lev) continue;
        if (once && !first && !(d >> 24)) { S(efc_jar)[k] = 0; continue; }
        const int a = d & 0xff, n = (d >> 8) & 0xff;
        const int tid = SI(efc_tid)[SI(ns_row)[a]], t = EFC_TYPE(tid), id = EFC_ID(tid);
        if (n == 1) noslip_block_lane<1>(a, t, id, k);
        else if (n == 2) noslip_block_lane<2>(a, t, id, k);
        else if (n == 3) noslip_block_lane<3>(a, t, id, k);
        else noslip_block_lane<5>(a, t, id, k);
      }
      DMC_WSYNC();
      for (int k = 0; k < nb; k++) {      // the rows OUTSIDE a block that its new forces move: only blocks with a partner
        const int d = SI(ns_blk)[k];
        if (((d >> 16) & 0xff) != lev || !(d >> 24)) continue;
        const int a = d & 0xff, n = (d >> 8) & 0xff;
        for (int p = 0; p < n; p++) {
          const T delta = S(efc_jv)[a + p];
          if (delta != 0) { const T* Ap = ns_A() + (a + p)*L.d.nslip; for (int b = lane; b < nf; b += LPE) if (b < a || b >= a + n) S(ns_res)[b] += Ap[b]*delta; }
          DMC_WSYNC();
        }
      }
    }
    T total = 0;
    for (int k = 0; k < nb; k++) total += S(efc_jar)[k];
    return total;
  }
  // A is symmetric and stored in full, (nslip, nslip) per environment in global memory: entry (i, j), i >= j, is
  // computed once as J_i . (M^-1 J_j^T) and written to both places, so that every later read runs along a row
  DMC_DEV T* ns_A() const { return (T*)o.ns_A + (size_t)SI(imisc)[IM_ENV] * L.d.nslip * L.d.nslip; }
#ifndef DMC_HOST_EMU
  // A = J_F M^-1 J_F^T for model-specialised kernels: lane i loads row i and column i of M's factor ONCE and keeps
  // them in registers over all nf solves; each solve is chol_solve_rows' two register sweeps, operation for operation.
  template <int N>
  DMC_DEV void noslip_build_A_rows(const DMC_LDS T* Lm, int nf, const RowMap& rm) {
    const int i = lane;
    const bool own = i < N;
    const int ci = tri_c0(own ? i : 0, N);
    T row[N], col[N];
#pragma unroll
    for (int k = 0; k < N; k++) { row[k] = (own && k < i) ? Lm[tri_c0(k, N) + i - k] : (T)0; col[k] = (own && k > i) ? Lm[ci + k - i] : (T)0; }
    const T dinv = own ? Lm[ci] : (T)0;
    T* A = ns_A(); const int cap = L.d.nslip;
    // M (and its factor) is block diagonal over the kinematic trees: a friction row only has entries in the dofs of its
    // contact's trees, so the substitutions only need the steps from the first of those dofs to the end of the last
    // tree (forward) and back to the start of the first (backward) -- the skipped steps would multiply exact zeros.
    // One environment per wave only (the bounds must be wave-uniform).
    int t_start = 0, t_end = N;
    if (LPE == 64 && L.d.ntree > 1 && own) {
      const unsigned lo = (unsigned)MI(dof_anc_lo)[i], hi = N > 32 ? (unsigned)MI(dof_anc_hi)[i] : 0u;
      t_start = lo ? __builtin_ctz(lo) : 32 + __builtin_ctz(hi);      // root dof of the lane's tree
      t_end = MI(dof_subend)[t_start];
    }
    for (int b = 0; b < nf; b++) {
      T sreg = own ? row_entry(SI(ns_row)[b], i, rm) : (T)0;
      int kf = 0, kb = 0, ke = N;
      if (LPE == 64 && L.d.ntree > 1) {
        const unsigned long long nz = __ballot(sreg != 0);
        if (nz) {
          const int first = __builtin_ctzll(nz), last = 63 - __builtin_clzll(nz);
          kf = first; kb = __builtin_amdgcn_readlane(t_start, first); ke = __builtin_amdgcn_readlane(t_end, last);
        } else { kf = N; kb = N; ke = 0; }
      }
      // (chol_solve_rows' three-instruction steps: one product per lane, one cross-lane read, one unconditional FMA; a lane
      // outside the swept range holds s = 0 throughout)
#pragma unroll
      for (int k = 0; k < N; k++) {
        if (k < kf || k >= ke) continue;
        const T xk = wave_bcast<LPE>(sreg*dinv, k);
        sreg = sreg - row[k]*xk;
      }
      sreg = sreg*dinv;
#pragma unroll
      for (int k = N - 1; k >= 0; k--) {
        if (k < kb || k >= ke) continue;
        const T xk = wave_bcast<LPE>(sreg*dinv, k);
        sreg = sreg - col[k]*xk;
      }
      if (own) S(sv_Mgrad)[i] = sreg*dinv;
      DMC_WSYNC();
      for (int a = b + lane; a < nf; a += LPE) { const T v = row_dot(SI(ns_row)[a], S(sv_Mgrad), rm); A[b*cap + a] = v; A[a*cap + b] = v; }
      DMC_WSYNC();
    }
  }
  // The same with one LANE PER FRICTION ROW (at most 64 rows, small nv): lane b keeps row b of J_F and solves
  // M x = J_b' in its own registers -- the entries of the factor are wave-uniform LDS reads, every FMA works for all rows
  // at once -- then A[a][b] = J_a . x_b with J_a's entries read across lanes.  The per-row version above is a chain of
  // 2 N dependent cross-lane steps plus a global-memory round trip PER ROW (12 % of the soccer step for ~10 rows); this
  // one is ~N^2 FMAs for all rows together.  Operation for operation the same sums (the skipped products are exact zeros).
  template <int N>
  DMC_DEV void noslip_build_A_lanes(const DMC_LDS T* Lm_, int nf, const RowMap& rm) {
    const int b = lane;
    const bool own = b < nf;
    const int rb = SI(ns_row)[own ? b : 0];
    T jd[N], x[N];
    // row b of J_F as N dense entries, branch-free (all loads in flight together): a friction row is a dof-friction row
    // (one nonzero) or a contact row (compressed to the dofs of its mask); all-dense models read their dense row
    if (L.d.jfull) {
#pragma unroll
      for (int k = 0; k < N; k++) jd[k] = own ? S(efc_Jd)[rb*N + k] : (T)0;
    } else {
      const int tid = SI(efc_tid)[rb];
      const bool con = rb >= rm.c0;
      const int c = con ? EFC_ID(tid) : 0;
      const unsigned mask = !own ? 0u : (con ? con_mask_lo(c) : (1u << simple_dof(tid)));
      const T one = con ? (T)0 : simple_sign(tid);
      const auto jr = Jc() + (con ? rb - rm.c0 : 0)*L.d.kmax;
      const int last = L.d.kmax - 1;
#pragma unroll
      for (int k = 0; k < N; k++) {
        const int sl = __builtin_popcount(mask & ((1u << k) - 1u));
        const T v = jr[sl < last ? sl : last];
        jd[k] = ((mask >> k) & 1u) ? (con ? v : one) : (T)0;
      }
    }
#pragma unroll
    for (int k = 0; k < N; k++) x[k] = jd[k];
    // The next column's entries are requested before the current one is used.  The loads go through a volatile pointer
    // and every updated value is pinned (an empty asm) at the end of its column: the updates carry no side effect, and
    // without the pins instruction selection emits all N^2 / 2 loads of a sweep first and the updates last -- the loaded
    // factor then lives in scratch memory.
#define DMC_PIN(v) asm volatile("" : "+v"(v))
    const volatile DMC_LDS T* Lm = Lm_;
    T cur[N], nxt[N];
    // Trees of equal size (StepDims::treeuni: soccer's five 6-dof trees): the factor of M is block diagonal with blocks
    // known at compile time, so column k only reaches to the end / row k only back to the start of its tree -- 2 x 105
    // loads and FMAs instead of 2 x 465 for N = 30; the skipped ones multiply exact zeros.
#ifdef DMC_NO_TREE_SPLIT
    constexpr int TB = N;
#else
    constexpr int TB = (LS::kTreeUni && LS::kTreeMax > 0) ? LS::kTreeMax : N;
#endif
#define DMC_TEND(k) ((((k) / TB) + 1) * TB < N ? (((k) / TB) + 1) * TB : N)
#define DMC_TBEG(k) (((k) / TB) * TB)
#pragma unroll
    for (int i = 0; i < N; i++) { cur[i] = i < DMC_TEND(0) ? Lm[i] : (T)0; nxt[i] = 0; }
#pragma unroll
    for (int k = 0; k < N; k++) {
      if (k + 1 < N) {
        const int cn = tri_c0(k + 1, N);
#pragma unroll
        for (int i = k + 1; i < DMC_TEND(k + 1); i++) nxt[i] = Lm[cn + i - (k + 1)];
      }
      const T xk = x[k] * cur[k];
      x[k] = xk;
#pragma unroll
      for (int i = k + 1; i < DMC_TEND(k); i++) { x[i] -= cur[i]*xk; DMC_PIN(x[i]); }
#pragma unroll
      for (int i = 0; i < N; i++) cur[i] = nxt[i];
    }
    // back substitution: step k needs 1 / L[k][k] and row k of L (entries (k, i), i < k)
#pragma unroll
    for (int i = DMC_TBEG(N - 1); i < N; i++) cur[i] = Lm[tri_c0(i, N) + (N - 1) - i];
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
      if (k > 0) {
#pragma unroll
        for (int i = DMC_TBEG(k - 1); i < k; i++) nxt[i] = Lm[tri_c0(i, N) + (k - 1) - i];
      }
      const T xk = x[k] * cur[k];
      x[k] = xk;
#pragma unroll
      for (int i = DMC_TBEG(k); i < k; i++) { x[i] -= cur[i]*xk; DMC_PIN(x[i]); }
#pragma unroll
      for (int i = 0; i < N; i++) cur[i] = nxt[i];
    }
#undef DMC_TEND
#undef DMC_TBEG
#undef DMC_PIN
    T* A = ns_A(); const int cap = L.d.nslip;
    for (int a = 0; a < nf; a++) {
      T v = 0;
#pragma unroll
      for (int k = 0; k < N; k++) v += wave_bcast<LPE>(jd[k], a) * x[k];
      if (own && a >= b) { A[b*cap + a] = v; A[a*cap + b] = v; }
    }
    DMC_WSYNC();
  }
#endif
  DMC_DEV void noslip(int nefc) {
    const int nv = L.d.nv, cap = L.d.nslip;
    int nf = 0, over = 0;
#if defined(DMC_HOST_EMU) || defined(DMC_NS_SCAN_SERIAL)
    for (int i = 0; i < nefc; i++) {
      const int tid = SI(efc_tid)[i], t = EFC_TYPE(tid);
      bool take = t == EFC_FRICTION || t == EFC_PYRAMIDAL;
      if (t == EFC_ELLIPTIC) take = i != SI(con_efc)[EFC_ID(tid)];
      if (take) { if (nf < cap) { if (lane == 0) SI(ns_row)[nf] = i; nf++; } else over = 1; }
    }
#else
    // the friction rows in row order: one lane per row and a prefix sum (the row-by-row walk was two dependent LDS round
    // trips per row for every lane: ~3 k cycles for the 15 rows of a soccer step)
    for (int i0 = 0; i0 < nefc; i0 += LPE) {
      const int i = i0 + lane, ii = i < nefc ? i : 0;
      const int tid = SI(efc_tid)[ii], t = EFC_TYPE(tid);
      const int r0 = SI(con_efc)[t == EFC_ELLIPTIC ? EFC_ID(tid) : 0];
      const int take = (i < nefc && (t == EFC_FRICTION || t == EFC_PYRAMIDAL || (t == EFC_ELLIPTIC && i != r0))) ? 1 : 0;
      int total;
      const int pos = nf + group_scan<LPE>(take, lane, &total);
      if (take && pos < cap) SI(ns_row)[pos] = i;
      nf += total;
    }
    if (nf > cap) over = 1;
#endif
    if (over) { if (lane == 0) SI(imisc)[IM_WARN + DMC_WARN_CNSTRFULL]++; return; }   // more friction rows than the cap: step without noslip
    if (!nf) return;
    // M^-1 through the factor of M computed in the position stage and kept beside H's factor: in qLM, or (large
    // models) in the global scratch, from where it returns to qLH now that the solver is done with H
    DMC_WSYNC();
    if (L.d.jglobal) {
      const DMC_GLB T* g = (const DMC_GLB T*)gLM();
      FOR_LANES(i, L.d.ntri) S(qLH)[i] = g[i];
      DMC_WSYNC();
    }
    const RowMap rm = row_map();
    bool built = false;
#ifndef DMC_HOST_EMU
    // (out of line: its own register allocation -- inside the acceleration stage its 2 N live values spilled)
    if constexpr (LS::kNV > 0 && LS::kNV <= 32 && LPE == 64) {
      if (nf <= LPE) {
        StageFns<T, LPE, LS>::ns_build(ls, (const DMC_LDS StepOpts<T>*)&o, (DMC_LDS int*)mi, (DMC_LDS T*)mr, gc, (DMC_LDS T*)s, (DMC_LDS int*)si, lane, nf);
        built = true;
      }
    }
    if constexpr (LS::kNV > 0 && LS::kNV <= LPE) { if (!built) { noslip_build_A_rows<LS::kNV>((const DMC_LDS T*)M_factor(), nf, rm); built = true; } }
#endif
    if (!built) for (int b = 0; b < nf; b++) {
      const int rb = SI(ns_row)[b];
      FOR_LANES(i, nv) S(sv_grad)[i] = row_entry(rb, i, rm);
      DMC_WSYNC();
      chol_solve(S(sv_Mgrad), M_factor(), S(sv_grad), nv, true);
      { T* A = ns_A(); const int cap = L.d.nslip;
        for (int a = b + lane; a < nf; a += LPE) { const T v = row_dot(SI(ns_row)[a], S(sv_Mgrad), rm); A[b*cap + a] = v; A[a*cap + b] = v; } }
      DMC_WSYNC();
    }
    DMC_PROF(PROF_X6);
    for (int a = lane; a < nf; a += LPE) { const int ra = SI(ns_row)[a]; S(ns_res)[a] = row_dot(ra, S(qacc), rm) - S(efc_aref)[ra]; }
    DMC_WSYNC();
    const T scale = 1 / (o.meaninertia * (T)(nv > 1 ? nv : 1));
    int nlev = 0;
    const int nblk = noslip_plan(nf, &nlev);
    int iter = 0;
    while (iter < o.noslip_iterations) {
      T improvement = 0;
      if (iter == 0) {
        for (int i = lane; i < nefc; i += LPE) { const T f = S(efc_force)[i]; improvement += (T)0.5*f*f / S(efc_D)[i]; }
        improvement = group_sum<LPE>(improvement);
      }
      if (nblk) improvement -= noslip_sweep_levels(nf, nblk, nlev, iter == 0);
      else for (int a = 0; a < nf; ) {
        const int i = SI(ns_row)[a], tid = SI(efc_tid)[i], t = EFC_TYPE(tid), id = EFC_ID(tid);
        int n = 1;
        if (t == EFC_PYRAMIDAL) n = 2;
        else if (t == EFC_ELLIPTIC) n = con_dim(id) - 1;
        T change;
        if (n == 1) change = noslip_block<1>(a, nf, t, id);
        else if (n == 2) change = noslip_block<2>(a, nf, t, id);
        else if (n == 3) change = noslip_block<3>(a, nf, t, id);
        else change = noslip_block<5>(a, nf, t, id);
        improvement -= change;
        a += n;
      }
      improvement *= scale;
#ifdef DMC_HOST_EMU
      if (getenv("DMC_EMU_TRACE_NS")) fprintf(stderr, "noslip sweep %d improvement %.6e (tol %.1e)\n", iter, (double)improvement, (double)o.noslip_tolerance);
#endif
      iter++;
      if (improvement < o.noslip_tolerance) break;
#if !defined(DMC_EXACT_QCQP) && !defined(DMC_NS_RESWEEP_ALL)
      // fp32 solves a block without a partner once (noslip_sweep_levels): when NO block has one -- one level: every
      // contact on a tree of its own, the usual state of the soccer pitch -- the later sweeps would skip every block and
      // add up zeros (noslip_tolerance = 0 never ends them: 0 < 0), so the pass ends here with the same forces
      if (sizeof(T) == 4 && nblk && nlev == 1) break;
#endif
    }
#ifdef DMC_HOST_EMU
    emu_ls_counts()[2]++; emu_ls_counts()[3] += iter;
#endif
    DMC_PROF(PROF_X7);
    constraint_force_to_joint(nefc);
    DMC_WSYNC();
    FOR_LANES(i, nv) S(sv_grad)[i] = S(qfrc_smooth)[i] + S(qfrc_constraint)[i];
    DMC_WSYNC();
    chol_solve(S(qacc), M_factor(), S(sv_grad), nv, true);
    DMC_WSYNC();
  }
  // ---- PGS (mj_solPGS; option solver="PGS", dm_control/mjcf/schema.xml:69-72; same algorithm and operation order as the
  // oracle's pgs_solve).  Dual problem  min 1/2 f'AR f + f'b,  AR = J M^-1 J' + diag(R),  b = J qacc_smooth - aref:
  // Gauss-Seidel over the rows in force space.  AR is built once per solve (one substitution per row, as noslip builds
  // its A) into the environment's (njmax, njmax) matrix in global memory; the residual  res = b + AR f  is kept up to
  // date lane-parallel (row i of the symmetric AR = column i, coalesced), the per-block scalar maths runs redundantly
  // on every lane.  A block update that raises the cost by more than 1e-10 is undone.
  template <int N>
  DMC_DEV T pgs_cone_block(int i) {
    const int cap = L.d.nslip, c = EFC_ID(SI(efc_tid)[i]);
    T A[N*N], old[N], rs[N], f[N];
#pragma unroll
    for (int p = 0; p < N; p++) {
      old[p] = S(efc_force)[i + p]; rs[p] = S(ns_res)[i + p]; f[p] = old[p];
#pragma unroll
      for (int q = 0; q < N; q++) A[p*N + q] = ns_A()[(i + p)*cap + i + q];
    }
    const T* fr3 = MR(prm_friction) + 3*con_prm(c);
    const T fri5[5] = {fr3[0], fr3[0], fr3[1], fr3[2], fr3[2]};
    if (f[0] < (T)DMC_MINVAL) {            // normal force too small: normal update, friction cleared
      f[0] -= t_div_exact(rs[0], A[0]);
      if (f[0] < 0) f[0] = 0;
#pragma unroll
      for (int p = 1; p < N; p++) f[p] = 0;
    } else {                               // ray update along the current force
      T v1[N], denom = 0, vr = 0;
#pragma unroll
      for (int p = 0; p < N; p++) { T t = 0;
#pragma unroll
        for (int q = 0; q < N; q++) t += A[p*N + q]*old[q];
        v1[p] = t; }
#pragma unroll
      for (int p = 0; p < N; p++) { denom += old[p]*v1[p]; vr += old[p]*rs[p]; }
      if (denom >= (T)DMC_MINVAL) {
        T x = t_div_exact(-vr, denom);
        if (f[0] + x*old[0] < 0) x = t_div_exact(-f[0], old[0]);
#pragma unroll
        for (int p = 0; p < N; p++) f[p] += x*old[p];
      }
    }
    // friction update with the normal fixed
    constexpr int M = N - 1;
    T Ac[M*M], bc[M], fri[M], v[M];
#pragma unroll
    for (int p = 0; p < M; p++) {
      fri[p] = fri5[p < 5 ? p : 4];
      bc[p] = rs[p + 1];
#pragma unroll
      for (int q = 0; q < M; q++) { Ac[p*M + q] = A[(p + 1)*N + q + 1]; bc[p] -= Ac[p*M + q]*old[1 + q]; }
      bc[p] += A[(p + 1)*N]*(f[0] - old[0]);
    }
    if (f[0] < (T)DMC_MINVAL) {
#pragma unroll
      for (int p = 0; p < M; p++) f[1 + p] = 0;
    } else {
      const int active = qcqp<M>(v, Ac, bc, fri, f[0]);
      if (active) {
        T ss = 0;
#pragma unroll
        for (int p = 0; p < M; p++) ss += t_div_exact(v[p]*v[p], fri[p]*fri[p]);
        ss = t_sqrt_exact(t_div_exact(f[0]*f[0], t_max((T)DMC_MINVAL, ss)));
#pragma unroll
        for (int p = 0; p < M; p++) v[p] *= ss;
      }
#pragma unroll
      for (int p = 0; p < M; p++) f[1 + p] = v[p];
    }
    T change = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
      T tq = 0;
#pragma unroll
      for (int q = 0; q < N; q++) tq += A[p*N + q]*(f[q] - old[q]);
      change += (T)0.5*(f[p] - old[p])*tq + (f[p] - old[p])*rs[p];
    }
    if (change > (T)1e-10) {
#pragma unroll
      for (int p = 0; p < N; p++) f[p] = old[p];
      change = 0;
    }
    DMC_WSYNC();
    const int nefc = SI(imisc)[IM_NEFC];
#pragma unroll
    for (int p = 0; p < N; p++) {
      const T delta = f[p] - old[p];
      if (lane == 0) S(efc_force)[i + p] = f[p];
      if (delta != 0) { const T* Ap = ns_A() + (i + p)*cap; for (int b = lane; b < nefc; b += LPE) S(ns_res)[b] += Ap[b]*delta; }
    }
    DMC_WSYNC();
    return change;
  }
  DMC_DEV void pgs_solve(int nefc) {
    const int nv = L.d.nv, cap = L.d.nslip;
    const RowMap rm = row_map();
    for (int i = lane; i < nefc; i += LPE) SI(ns_row)[i] = i;
    DMC_WSYNC();
    // AR = J M^-1 J' + diag(R)
    bool built = false;
#ifndef DMC_HOST_EMU
    if constexpr (LS::kNV > 0 && LS::kNV <= LPE) { noslip_build_A_rows<LS::kNV>((const DMC_LDS T*)M_factor(), nefc, rm); built = true; }
#endif
    if (!built) for (int b = 0; b < nefc; b++) {
      FOR_LANES(i, nv) S(sv_grad)[i] = row_entry(b, i, rm);
      DMC_WSYNC();
      chol_solve(S(sv_Mgrad), M_factor(), S(sv_grad), nv, true);
      { T* A = ns_A();
        for (int a = b + lane; a < nefc; a += LPE) { const T v = row_dot(a, S(sv_Mgrad), rm); A[b*cap + a] = v; A[a*cap + b] = v; } }
      DMC_WSYNC();
    }
    for (int i = lane; i < nefc; i += LPE) {
      ns_A()[i*cap + i] += t_div_exact((T)1, S(efc_D)[i]);
      S(efc_jv)[i] = row_dot(i, S(qacc_smooth), rm) - S(efc_aref)[i];       // b
    }
    // warm start: the forces of qacc_warmstart through the primal map, kept only if their dual cost beats zero forces
    bool warm = false;
    if (!(o.disableflags & DMC_DSBL_WARMSTART)) {
      for (int i = lane; i < nefc; i += LPE) S(efc_jar)[i] = row_dot(i, S(qacc_warmstart), rm) - S(efc_aref)[i];
      DMC_WSYNC();
      constraint_update(nefc);
      T cost = 0;
      for (int i = lane; i < nefc; i += LPE) {
        const T* Ai = ns_A() + i*cap;
        T t = 0;
        for (int j = 0; j < nefc; j++) t += Ai[j]*S(efc_force)[j];
        const T f = S(efc_force)[i], bi = S(efc_jv)[i];
        cost += f*bi + (T)0.5*f*t;
        S(ns_res)[i] = bi + t;
      }
      cost = group_sum<LPE>(cost);
      warm = !(cost > 0);
    }
    DMC_WSYNC();
    if (!warm) for (int i = lane; i < nefc; i += LPE) { S(efc_force)[i] = 0; S(ns_res)[i] = S(efc_jv)[i]; }
    DMC_WSYNC();
    const T scale = 1 / (o.meaninertia * (T)(nv > 1 ? nv : 1));
    int iter = 0;
    while (iter < o.iterations) {
      T improvement = 0;
      for (int i = 0; i < nefc; ) {
        const int tid = SI(efc_tid)[i], t = EFC_TYPE(tid);
        const int dim = t == EFC_ELLIPTIC ? con_dim(EFC_ID(tid)) : 1;
        if (dim == 1) {
          const T a = ns_A()[i*cap + i], r = S(ns_res)[i], old = S(efc_force)[i];
          T f = old - t_div_exact(r, a);
          if (t == EFC_FRICTION) { const T fl = MR(dof_frictionloss)[EFC_ID(tid)]; if (f < -fl) f = -fl; else if (f > fl) f = fl; }
          else if (t != EFC_EQUALITY) { if (f < 0) f = 0; }
          T delta = f - old;
          T change = (T)0.5*delta*delta*a + delta*r;
          if (change > (T)1e-10) { f = old; delta = 0; change = 0; }
          DMC_WSYNC();
          if (lane == 0) S(efc_force)[i] = f;
          if (delta != 0) { const T* Ap = ns_A() + i*cap; for (int b = lane; b < nefc; b += LPE) S(ns_res)[b] += Ap[b]*delta; }
          DMC_WSYNC();
          improvement -= change;
        } else if (dim == 3) improvement -= pgs_cone_block<3>(i);
        else if (dim == 4) improvement -= pgs_cone_block<4>(i);
        else improvement -= pgs_cone_block<6>(i);
        i += dim;
      }
      improvement *= scale;
      iter++;
      if (improvement < o.tolerance) break;
    }
    // dualFinish: qfrc_constraint = J' f, qacc = qacc_smooth + M^-1 qfrc_constraint
    constraint_force_to_joint(nefc);
    DMC_WSYNC();
    chol_solve(S(sv_Mgrad), M_factor(), S(qfrc_constraint), nv, true);
    FOR_LANES(i, nv) { const T a = S(qacc_smooth)[i] + S(sv_Mgrad)[i]; S(qacc)[i] = a; S(qacc_warmstart)[i] = a; }
    if (lane == 0) SI(imisc)[IM_ITER] = iter;
    DMC_WSYNC();
  }
  DMC_DEV void fwd_constraint() {
    const int nv = L.d.nv, nefc = SI(imisc)[IM_NEFC];
    if (!nefc) {
      FOR_LANES(i, nv) { const T a = S(qacc_smooth)[i]; S(qacc)[i] = a; S(qacc_warmstart)[i] = a; S(qfrc_constraint)[i] = 0; }
      if (lane == 0) SI(imisc)[IM_ITER] = 0;
      DMC_WSYNC();
      return;
    }
    if (L.d.pgs) {
      pgs_solve(nefc);
      if (L.d.nslip) { if (o.noslip_iterations > 0) noslip(nefc); }
      return;
    }
    // Warm start (mj_solPrimal / warmstart()): keep qacc_warmstart only if its cost beats qacc_smooth's.  qacc_smooth is
    // evaluated first, so that when the warm start wins -- the usual case -- its M a, J a - aref, forces, active set and
    // cost are already those the solver starts from (they used to be recomputed: a third of the solver's set-up).
    int changed = 1;
    T cc = 0, gauss = 0;
    bool evaluated = false;
    if (!(o.disableflags & DMC_DSBL_WARMSTART)) {
      jar_from(S(qacc_smooth), nefc);
      DMC_WSYNC();
      const T cs = constraint_update(nefc);      // Gauss term is zero at qacc_smooth
      FOR_LANES(i, nv) S(qacc)[i] = S(qacc_warmstart)[i];
      DMC_WSYNC();
      jar_from(S(qacc), nefc);
      mul_M(S(sv_Ma), S(qacc));
      DMC_WSYNC();
      for (int i = lane; i < nefc; i += LPE) SI(efc_active)[i] = -1;   // no factor of H yet
      cc = constraint_update(nefc, &changed);
      gauss = gauss_cost();
      if (cc + gauss > cs) FOR_LANES(i, nv) S(qacc)[i] = S(qacc_smooth)[i];
      else evaluated = true;
    } else FOR_LANES(i, nv) S(qacc)[i] = S(qacc_smooth)[i];
    DMC_WSYNC();
    int iter;
    // islands -1 (default): per-island solves where they change the answer measurably -- fp64 batches (held to the CPU
    // reference) and CG at any precision (CG stops at a looser point: joint vs per-island answers 4.9e-6 apart); fp32
    // Newton solves jointly (8e-15 apart, and a solve per island on one wave saves nothing)
    if (L.d.island && (o.islands < 0 ? (sizeof(T) == 8 || L.d.cg) : o.islands != 0) && !(o.disableflags & DMC_DSBL_ISLAND) && solve_islands(nefc, &iter)) constraint_force_to_joint(nefc);
    else {
      iter = primal_solve(nefc, evaluated, cc, gauss, changed);
      // (qfrc_constraint = J' efc_force is already that of the solution: every exit of primal_solve is preceded by a
      // newton_gradient at the final (qacc, efc_force), whose first act is this product -- forming it again was 5 % of the
      // 9-dof step)
    }
    FOR_LANES(i, nv) S(qacc_warmstart)[i] = S(qacc)[i];
    if (lane == 0) SI(imisc)[IM_ITER] = iter;
    DMC_WSYNC();
    // the warm start keeps the main solver's solution; noslip then edits qacc / efc_force
    DMC_PROF(PROF_SOL_UPD);
    if (L.d.nslip) { if (o.noslip_iterations > 0) noslip(nefc); }
    DMC_PROF(PROF_NOSLIP);
  }
  // mj_solPrimal (Newton / CG) from S(qacc) over every dof and row of the environment; `evaluated`: M a, J a - aref, the
  // forces, the active set and the cost (cc, gauss, changed) of the starting point are already in place
  DMC_DEV int primal_solve(int nefc, bool evaluated, T cc, T gauss, int changed) {
    const int nv = L.d.nv;
    const T scale = 1 / (o.meaninertia * (T)(nv > 1 ? nv : 1));
    if (!evaluated) {
      mul_M(S(sv_Ma), S(qacc));
      jar_from(S(qacc), nefc);
      DMC_WSYNC();
      changed = 1;
      for (int i = lane; i < nefc; i += LPE) SI(efc_active)[i] = -1;   // no factor of H yet
      cc = constraint_update(nefc, &changed);
      gauss = gauss_cost();
    }
    T cost = cc + gauss;
    // (device: only the kernels that have the side-by-side routines leave H's other entries unwritten)
#ifdef DMC_HOST_EMU
    hsplit = !L.d.cg && h_split(nefc);
    emu_split_solves() += hsplit ? 1 : 0;
#else
    hsplit = false;
    if constexpr (kSplit) hsplit = !L.d.cg && h_split(nefc);
#endif
#ifdef DMC_HOST_EMU
    if (getenv("DMC_EMU_TRACE_SPLIT")) fprintf(stderr, "h_split %d (treemax %d, ncon %d)\n", (int)hsplit, L.d.treemax, SI(imisc)[IM_NCON]);
#endif
    DMC_PROF(PROF_SOL_INIT);
    newton_gradient(nefc, 1);
    FOR_LANES(i, nv) S(sv_search)[i] = -S(sv_Mgrad)[i];
    DMC_WSYNC();
    DMC_PROF(PROF_SOL_GRAD);
    int iter = 0, stall = 0;
    T gbest = (T)DMC_MAXVAL;
#ifndef DMC_STALL_ITERS
#define DMC_STALL_ITERS 6
#endif
    while (iter < o.iterations) {
#ifndef DMC_HOST_EMU
      // a launch ends with its slowest wave, and that is one whose solve takes many iterations: from the third
      // iteration on it wins the issue arbitration against the wave it shares the SIMD with (which has slack)
      if (iter == DMC_PRIO_ITER) __builtin_amdgcn_s_setprio(2);
#endif
      T lscost;
      DMC_TSUB(3, iter == 0, 4);
      const T alpha = primal_search(nefc, gauss, scale, &lscost);
      DMC_PROF(PROF_SOL_LS);
      DMC_TSUB(3, iter == 0, 5);
#ifdef DMC_HOST_EMU
      if (getenv("DMC_EMU_TRACE_LS")) fprintf(stderr, "    ls returned alpha %.9e lscost %.6e\n", (double)alpha, (double)lscost);
#endif
      if (alpha == 0) break;
      FOR_LANES(i, nv) { S(qacc)[i] += alpha*S(sv_search)[i]; S(sv_Ma)[i] += alpha*S(sv_Mv)[i]; }
      for (int i = lane; i < nefc; i += LPE) S(efc_jar)[i] += alpha*S(efc_jv)[i];
      DMC_WSYNC();
      const T oldcost = cost;
      if (L.d.cg) FOR_LANES(i, nv) { S(sv_gold)[i] = S(sv_grad)[i]; S(sv_Mgold)[i] = S(sv_Mgrad)[i]; }
      cc = constraint_update(nefc, &changed);
      gauss = gauss_cost();
      cost = cc + gauss;
      DMC_PROF(PROF_SOL_UPD);
      DMC_TSUB(3, iter == 0, 6);
      newton_gradient(nefc, changed);
      DMC_PROF(PROF_SOL_GRAD);
      DMC_TSUB(3, iter == 0, 7);
      T beta = 0;
      if (L.d.cg) {      // Polak-Ribiere, restarted when negative
        T num = 0, den = 0;
        FOR_LANES(i, nv) { num += S(sv_grad)[i]*(S(sv_Mgrad)[i] - S(sv_Mgold)[i]); den += S(sv_gold)[i]*S(sv_Mgold)[i]; }
        num = group_sum<LPE>(num); den = group_sum<LPE>(den);
        beta = num / t_max((T)DMC_MINVAL, den);
        if (beta < 0) beta = 0;
      }
      T g2 = 0, ma2 = 0;
      FOR_LANES(i, nv) {
        T sr = -S(sv_Mgrad)[i];
        if (L.d.cg) sr += beta*S(sv_search)[i];
        S(sv_search)[i] = sr; g2 += S(sv_grad)[i]*S(sv_grad)[i]; ma2 += S(sv_Ma)[i]*S(sv_Ma)[i];
      }
      g2 = group_sum<LPE>(g2); ma2 = group_sum<LPE>(ma2);
      DMC_WSYNC();
      const T improvement = ls_relative<T>() ? -scale*lscost : scale*(oldcost - cost), gradient = scale*t_sqrt(g2);
      iter++;
      // MuJoCo's criteria, floored at what the arithmetic can resolve: a gradient below ~8 ulp of |M a| is rounding
      // noise, and so is a cost change below 1 ulp of the cost.  (Round 2 floored the improvement at 8 ulp: an iteration
      // whose line search stops at a kink makes a small but real step -- improvement 1e-3 against a cost of 7e5 --
      // while the gradient is still 1e4 x its floor; stopping there left one-step errors of 2e-4 on the 62-dof model,
      // 4e-8 with the 1-ulp floor, at no measurable cost: profiles/r03_fp32_floor.json.)  fp64: both floors are far
      // below `tolerance` (no-op).
#ifndef DMC_EPS_IMP
#define DMC_EPS_IMP 1
#endif
#ifdef DMC_HOST_EMU
      const T epsimp = getenv("DMC_EMU_EPSMUL") ? (T)atof(getenv("DMC_EMU_EPSMUL")) : (T)DMC_EPS_IMP;
#else
      const T epsimp = (T)DMC_EPS_IMP;
#endif
      const T ulp = sizeof(T) == 4 ? (T)1.1920929e-7 : (T)2.220446049250313e-16;
      const T tol_imp = ls_relative<T>() ? o.tolerance : t_max(o.tolerance, epsimp*ulp*scale*t_abs(cost));
      // fp32, round 5: an EXACT Newton step.  With every row in a quadratic / linear zone, an unchanged active set and the line search
      // returning the Newton point (alpha = 1), the step minimised the quadratic it was computed from: the new gradient is
      // zero in exact arithmetic (the fp64 solver sees ~1e-14 and stops here) and what fp32 shows -- measured 1x - 7x the
      // 8-ulp floor on the cheetah -- is the rounding of M a - qfrc_smooth - J' f itself.  Iterating on it moves qacc by
      // less than that noise (improvement ~1e-10) and cost the 9-dof model 13 % more Newton iterations than the fp64
      // reference (1.34 vs 1.18 per step; a launch waits for the wave with the most).  The floor is 64 ulp there.
      // Where: the 9-dof model (-14 % iterations, +0.9 % per single-step launch -- which waits for its slowest wave -- +3 % in
      // rollout mode; one-step error unchanged at 4.9e-7 max).  On the 27-dof humanoid it bought 1.5 % for a 5 x larger
      // one-step maximum (8.2e-7 -> 4.0e-6) and on the 62-dof walker nothing for 8.8e-6 -> 4.3e-5
      // (profiles/r05_ab_variants.log): models with nv <= 16 only.
      const bool exact_step = sizeof(T) == 4 && !L.d.cg && L.d.nv <= 16 && !changed && t_abs(alpha - 1) < (T)1e-3;      // (`changed` is also set by any contact in the cone's middle zone, whose cost is not quadratic)
      const T tol_grad = t_max(o.tolerance, (exact_step ? 64 : 8)*ulp*scale*t_sqrt(ma2));
#ifdef DMC_HOST_EMU
      if (getenv("DMC_EMU_TRACE")) fprintf(stderr, "  newton iter %d alpha %.6e cost %.9e improvement %.3e (tol %.3e) gradient %.3e (tol %.3e) changed %d\n",
                                            iter, (double)alpha, (double)cost, (double)improvement, (double)tol_imp, (double)gradient, (double)tol_grad, changed);
#endif
      if (improvement < tol_imp || gradient < tol_grad) break;
      // fp32 only: a solve that has stopped making progress is at the resolution of its arithmetic -- the gradient of a
      // 62-dof system stalls at 10-30 ulp of |M a|, above the 8-ulp floor, while the relative line search keeps finding
      // improvements of a few 1e-8: such a solve ran into the iteration cap (config 4: 0.02 % of the solves, 100
      // iterations against a mean of 3.4 and a 99.9th percentile of 11, profiles/r04_iter_hist_cfg4.json -- and a launch
      // waits for its longest item).  It ends when the gradient has not fallen by 10 % in six consecutive iterations.
      if (sizeof(T) == 4 && !L.d.cg && L.d.nv > 32) {      // (Newton only: CG converges linearly and may legitimately crawl; the stall was only ever seen on the 62-dof models, and the small kernels have no registers to spare)
        if (gradient < (T)0.9*gbest) { gbest = gradient; stall = 0; }
        else if (++stall >= DMC_STALL_ITERS && iter >= 2*DMC_STALL_ITERS) break;
      }
    }
#ifndef DMC_HOST_EMU
    __builtin_amdgcn_s_setprio(0);
#endif
    return iter;
  }
  // ---- constraint islands (mj_island + the per-island solves of mj_fwdConstraint; the oracle's find_islands /
  // solve_islands) -------------------------------------------------------------------------------------------------
  // Trees are named by their ROOT DOF (the lowest dof of the ancestor mask; nv <= 64, so a set of trees is a 64-bit
  // mask); every constraint row joins the trees whose dofs it moves; the islands are the connected components that own
  // a row.  Island k is then solved with the SAME joint solver: the dofs outside it are parked at qacc_smooth (their
  // block of M is independent, so their gradient is rounding noise) and the rows outside it are switched off by scaling
  // their D by a power of two far below the working precision (exactly undone afterwards) -- the island's iterates are
  // those of its own sub-problem to that precision, with its own line search, iteration count and stopping test.
  // One island that holds every dof and row is the joint problem: solved in place (bit-identical to the flag disabled).
  typedef unsigned long long u64;
#ifdef DMC_HOST_EMU
#define DMC_ATOMIC_OR(p, v) (*(p) |= (v))
#else
#define DMC_ATOMIC_OR(p, v) atomicOr((p), (v))
#endif
  DMC_DEV static int ctz64(u64 x) { return (unsigned)x ? __builtin_ctz((unsigned)x) : 32 + __builtin_ctz((unsigned)(x >> 32)); }
  DMC_DEV int dof_root(int i) const { return ctz64((u64)(unsigned)MI(dof_anc_lo)[i] | (u64)(unsigned)MI(dof_anc_hi)[i] << 32); }
  DMC_DEV u64 isl_comp(int t) const { return (u64)(unsigned)SI(isl_comp)[2*t] | (u64)(unsigned)SI(isl_comp)[2*t + 1] << 32; }
  DMC_DEV u64 row_trees(int r) const { return (u64)(unsigned)SI(efc_tree)[2*r] | (u64)(unsigned)SI(efc_tree)[2*r + 1] << 32; }
  DMC_DEV bool solve_islands(int nefc, int* iter_out) {
    const int nv = L.d.nv;
    const RowMap rm = row_map();
    FOR_LANES(i, nv) { const bool root = dof_root(i) == i; SI(isl_comp)[2*i] = (root && i < 32) ? (int)(1u << i) : 0; SI(isl_comp)[2*i + 1] = (root && i >= 32) ? (int)(1u << (i - 32)) : 0; }
    DMC_WSYNC();
    for (int r = lane; r < nefc; r += LPE) {
      u64 tm = 0;
      for (int i = 0; i < nv; i++) if (row_entry(r, i, rm) != 0) tm |= (u64)1 << dof_root(i);
      SI(efc_tree)[2*r] = (int)(unsigned)tm; SI(efc_tree)[2*r + 1] = (int)(unsigned)(tm >> 32);
    }
    DMC_WSYNC();
    // the rows of one contact travel together (a tangential row may move nothing on its own): every contact row takes
    // the union over its contact's rows (in place: the masks only grow towards that union)
    for (int r = lane; r < nefc; r += LPE) {
      const int tid = SI(efc_tid)[r], ty = EFC_TYPE(tid);
      if (ty != EFC_FRICTIONLESS && ty != EFC_PYRAMIDAL && ty != EFC_ELLIPTIC) continue;
      const int c = EFC_ID(tid), r0 = SI(con_efc)[c], nr = contact_rows(con_dim(c));
      u64 tm = 0;
      for (int q = r0; q < r0 + nr && q < nefc; q++) tm |= row_trees(q);
      SI(efc_tree)[2*r] = (int)(unsigned)tm; SI(efc_tree)[2*r + 1] = (int)(unsigned)(tm >> 32);
    }
    DMC_WSYNC();
    // every row joins the trees of its (unioned) mask -- AFTER the union: the normal row of an elliptic contact may move
    // only tree B and a tangent row only tree A; mj_island unites all dofs over the whole row group of the contact
    // (joined before the union, A and B stayed two components that both owned every row of the contact: each was then
    // solved as the joint problem -- right answer, twice the work)
    for (int r = lane; r < nefc; r += LPE) {
      const u64 tm = row_trees(r);
      for (u64 m = tm; m; m &= m - 1) { const int t = ctz64(m); DMC_ATOMIC_OR(&SI(isl_comp)[2*t], (int)(unsigned)tm); DMC_ATOMIC_OR(&SI(isl_comp)[2*t + 1], (int)(unsigned)(tm >> 32)); }
    }
    DMC_WSYNC();
    // transitive closure: every pass at least doubles the path length covered (6 passes: 64 trees).  In place: the masks only
    // grow towards the closure, so a lane reading another tree's half-updated mask still reads a subset of the answer
    for (int pass = 0; pass < 6; pass++) {
      FOR_LANES(t, nv) {
        const u64 c = isl_comp(t);
        u64 x = c;
        for (u64 m = c; m; m &= m - 1) x |= isl_comp(ctz64(m));
        SI(isl_comp)[2*t] = (int)(unsigned)x; SI(isl_comp)[2*t + 1] = (int)(unsigned)(x >> 32);
      }
      DMC_WSYNC();
    }
    // the islands: components that own a row, named by their lowest tree
    u64 rowtrees = 0, roots = 0;
    for (int r = 0; r < nefc; r++) rowtrees |= row_trees(r);
    for (int t = 0; t < nv; t++) if (isl_comp(t)) roots |= (u64)1 << t;
    int nisl = 0; u64 covered = 0;
    for (int t = 0; t < nv; t++) { const u64 C = isl_comp(t); if (C && ctz64(C) == t && (C & rowtrees)) { nisl++; covered |= C; } }
    bool whole = nisl == 1 && covered == roots;
    if (whole) for (int r = 0; r < nefc; r++) if (!row_trees(r)) whole = false;
#ifdef DMC_HOST_EMU
    if (getenv("DMC_EMU_TRACE")) { fprintf(stderr, "  islands: nefc %d nisl %d whole %d covered %llx roots %llx rowtrees %llx rows:", nefc, nisl, (int)whole, covered, roots, rowtrees); for (int r = 0; r < nefc; r++) fprintf(stderr, " %llx", row_trees(r)); fprintf(stderr, "\n"); }
#endif
    if (nisl == 0 || whole) return false;
    const T eps = sizeof(T) == 8 ? (T)8.271806125530277e-25 : (T)9.094947017729282e-13;      // 2^-80 / 2^-40
    FOR_LANES(i, nv) S(qacc_warmstart)[i] = S(qacc)[i];      // the start values (the warm-start choice has been made; the array is rewritten at the end)
    DMC_WSYNC();
    int itmax = 0;
    for (int t = 0; t < nv; t++) {
      const u64 C = isl_comp(t);
      if (!C || ctz64(C) != t || !(C & rowtrees)) continue;
      FOR_LANES(i, nv) S(qacc)[i] = ((C >> dof_root(i)) & 1) ? S(qacc_warmstart)[i] : S(qacc_smooth)[i];
      for (int r = lane; r < nefc; r += LPE) if (!(row_trees(r) & C)) S(efc_D)[r] *= eps;
      DMC_WSYNC();
      const int it = primal_solve(nefc, false, 0, 0, 1);
      FOR_LANES(i, nv) if ((C >> dof_root(i)) & 1) S(qacc_warmstart)[i] = S(qacc)[i];
      for (int r = lane; r < nefc; r += LPE) if (!(row_trees(r) & C)) S(efc_D)[r] *= 1 / eps;
      DMC_WSYNC();
      if (it > itmax) itmax = it;
    }
    // trees without a constraint take qacc_smooth; forces and states of every row at the assembled solution
    FOR_LANES(i, nv) S(qacc)[i] = ((covered >> dof_root(i)) & 1) ? S(qacc_warmstart)[i] : S(qacc_smooth)[i];
    DMC_WSYNC();
    jar_from(S(qacc), nefc);
    DMC_WSYNC();
    constraint_update(nefc);
    DMC_WSYNC();
    *iter_out = itmax;
    return true;
  }

  // ---- integration (mj_Euler with implicit joint damping) ---------------------------------
  DMC_DEV void euler_state() {
    const int nv = L.d.nv;
    const T dt = o.timestep;
    const T* qacc = S(qacc);
    const bool implicitfast = kFeat && o.integrator == DMC_INT_IMPLICITFAST;
    if (implicitfast) {
      // mj_implicit, mjINT_IMPLICITFAST: (M - h dF/dv) qacc = qfrc_smooth + qfrc_constraint with the velocity
      // derivatives of the passive and actuator forces and no Coriolis term.  The supported model class (the MJCF
      // compiler and make_dims refuse damped tendons, velocity-dependent tendon actuators and fluid forces with this
      // integrator) leaves dF/dv DIAGONAL: -dof_damping[i] (mjd_passive_vel) + gear^2 (biasprm[2] + gainprm[2] u) of
      // every actuator on dof i whose force is strictly inside its forcerange (mjd_actuator_vel).  The matrix is
      // always factored: mj_Euler's "any damping" shortcut and mjDSBL_EULERDAMP do not apply.
      // (evaluated before the activations advance: u is the activation the force was computed from)
      FOR_LANES(i, nv) {
        T dd = (o.disableflags & DMC_DSBL_DAMPER) ? (T)0 : MR(dof_damping)[i];
        if (!(o.disableflags & DMC_DSBL_ACTUATION)) for (int a = 0; a < L.d.nu; a++) {
          const int fl = MI(act_flags)[a];
          if ((fl & ACTF_TENDON) || MI(act_dof)[a] != i) continue;
          if (fl & ACTF_FORCELIMITED) { const T f = S(actuator_force)[a]; if (f <= MRC(act_forcerange)[2*a] || f >= MRC(act_forcerange)[2*a + 1]) continue; }
          T bv = (fl & ACTF_BIAS_AFFINE) ? MRC(act_biasprm)[3*a + 2] : (T)0;
          const T gv = (fl & ACTF_GAIN_AFFINE) ? MRC(act_gainprm)[3*a + 2] : (T)0;
          if (gv != 0) bv += gv * ((L.d.na && (fl & ACTF_DYN_ANY)) ? S(act)[MI(act_adr)[a]] : S(ctrl)[a]);
          const T gear = MRC(act_gear)[a];
          dd -= gear*gear*bv;
        }
        S(sv_search)[i] = dd;
      }
      DMC_WSYNC();
    }
    // activations first (mj_advance): explicit Euler, or the exact exponential for filterexact
    if (L.d.na) FOR_LANES(i, L.d.nu) {
      const int fl = MI(act_flags)[i];
      if (fl & ACTF_DYN_ANY) {
        const int k = MI(act_adr)[i];
        if (fl & ACTF_DYN_FILTEREXACT) { const T tau = t_max((T)DMC_MINVAL, MR(act_dynprm)[i]); S(act)[k] += S(act_dot)[k] * tau * (1 - t_exp(-dt/tau)); }
        else S(act)[k] += dt*S(act_dot)[k];
        if (fl & ACTF_ACTLIMITED) S(act)[k] = t_max(MRC(act_actrange)[2*i], t_min(MRC(act_actrange)[2*i + 1], S(act)[k]));   // mj_nextActivation
      }
    }
    if (implicitfast || (o.any_damping && !(o.disableflags & (DMC_DSBL_EULERDAMP | DMC_DSBL_DAMPER)))) {
      FOR_LANES(i, nv) S(sv_grad)[i] = S(qfrc_smooth)[i] + S(qfrc_constraint)[i];
      DMC_WSYNC();
      factor_M(true, implicitfast ? S(sv_search) : MR(dof_damping));
      chol_solve(S(sv_Mgrad), S(qLH), S(sv_grad), nv, true);
      qacc = S(sv_Mgrad);
    }
    FOR_LANES(i, nv) S(qvel)[i] += dt*qacc[i];
    DMC_WSYNC();
    FOR_LANES(j, L.d.njnt) {
      const int qa = MI(jnt_qposadr)[j], da = MI(jnt_dofadr)[j], t = MI(jnt_type)[j];
      if (t == DMC_JNT_FREE) {
        for (int k = 0; k < 3; k++) S(qpos)[qa + k] += dt*S(qvel)[da + k];
        T q[4], w[3];
        for (int k = 0; k < 4; k++) q[k] = S(qpos)[qa + 3 + k];
        for (int k = 0; k < 3; k++) w[k] = S(qvel)[da + 3 + k];
        quat_integrate(q, w, dt);
        for (int k = 0; k < 4; k++) S(qpos)[qa + 3 + k] = q[k];
      } else if (t == DMC_JNT_BALL) {
        T q[4], w[3];
        for (int k = 0; k < 4; k++) q[k] = S(qpos)[qa + k];
        for (int k = 0; k < 3; k++) w[k] = S(qvel)[da + k];
        quat_integrate(q, w, dt);
        for (int k = 0; k < 4; k++) S(qpos)[qa + k] = q[k];
      } else S(qpos)[qa] += dt*S(qvel)[da];
    }
    DMC_WSYNC();
  }
  DMC_DEV void euler() { euler_state(); time_ += o.timestep_d; }

  // ---- RK4 (mj_RungeKutta, N = 4): stages run through the single forward() site ----------
  DMC_DEV void integrate_pos_from(const T* q0, const T* vel, T h) {
    FOR_LANES(j, L.d.njnt) {
      const int qa = MI(jnt_qposadr)[j], da = MI(jnt_dofadr)[j], t = MI(jnt_type)[j];
      if (t == DMC_JNT_FREE || t == DMC_JNT_BALL) {
        int qo = qa, dofo = da;
        if (t == DMC_JNT_FREE) { for (int k = 0; k < 3; k++) S(qpos)[qa + k] = q0[qa + k] + h*vel[da + k]; qo += 3; dofo += 3; }
        T q[4], w[3];
        for (int k = 0; k < 4; k++) q[k] = q0[qo + k];
        for (int k = 0; k < 3; k++) w[k] = vel[dofo + k];
        quat_integrate(q, w, h);
        for (int k = 0; k < 4; k++) S(qpos)[qo + k] = q[k];
      } else S(qpos)[qa] = q0[qa] + h*vel[da];
    }
  }
  // after the forward pass of `stage`: accumulate B[stage]*F[stage]; prepare X[stage+1]
  DMC_DEV void rk4_stage(int stage) {
    const int nv = L.d.nv;
    const T h = o.timestep;
    const T Bw = (stage == 0 || stage == 3) ? (T)(1.0/6) : (T)(1.0/3);
    if (stage == 0) {
      FOR_LANES(i, L.d.nq) S(rk_q0)[i] = S(qpos)[i];
      FOR_LANES(i, nv) { S(rk_v0)[i] = S(qvel)[i]; S(rk_dq)[i] = Bw*S(qvel)[i]; S(rk_dv)[i] = Bw*S(qacc)[i]; }
    } else FOR_LANES(i, nv) { S(rk_dq)[i] += Bw*S(qvel)[i]; S(rk_dv)[i] += Bw*S(qacc)[i]; }
    DMC_WSYNC();
    if (stage < 3) {
      const T a = stage == 2 ? (T)1 : (T)0.5;
      // X[stage+1] = X[0] (+) h * a * F[stage]; sv_grad/sv_Mgrad are free scratch here
      FOR_LANES(i, nv) { S(sv_grad)[i] = a*S(qvel)[i]; S(sv_Mgrad)[i] = a*S(qacc)[i]; }
      DMC_WSYNC();
      integrate_pos_from(S(rk_q0), S(sv_grad), h);
      FOR_LANES(i, nv) S(qvel)[i] = S(rk_v0)[i] + h*S(sv_Mgrad)[i];
      DMC_WSYNC();
    }
  }
  DMC_DEV void rk4_finish() {
    const T h = o.timestep;
    FOR_LANES(i, L.d.nv) S(qvel)[i] = S(rk_v0)[i] + h*S(rk_dv)[i];
    integrate_pos_from(S(rk_q0), S(rk_dq), h);
    time_ += o.timestep_d;
    DMC_WSYNC();
  }

  // ---- checks / reset ---------------------------------------------------------------------
  DMC_DEV void reset_state() {
    FOR_LANES(i, L.d.nq) S(qpos)[i] = MR(qpos0)[i];
    FOR_LANES(i, L.d.nv) { S(qvel)[i] = 0; S(qacc_warmstart)[i] = 0; S(qfrc_applied)[i] = 0; }
    FOR_LANES(i, L.d.nu) S(ctrl)[i] = 0;
    if (L.d.na) FOR_LANES(i, L.d.na) { S(act)[i] = 0; S(act_dot)[i] = 0; }
    time_ = 0;
    DMC_WSYNC();
  }
  DMC_DEV bool check_pos_vel() {
    int badp = 0, badv = 0;
    FOR_LANES(i, L.d.nq) if (t_bad(S(qpos)[i])) badp = 1;
    badp = group_max<LPE>(badp);
    if (badp) { if (lane == 0) SI(imisc)[IM_WARN + DMC_WARN_BADQPOS]++; if (!(o.disableflags & DMC_DSBL_AUTORESET)) { DMC_WSYNC(); reset_state(); } }
    FOR_LANES(i, L.d.nv) if (t_bad(S(qvel)[i])) badv = 1;
    badv = group_max<LPE>(badv);
    if (badv) { if (lane == 0) SI(imisc)[IM_WARN + DMC_WARN_BADQVEL]++; if (!(o.disableflags & DMC_DSBL_AUTORESET)) { DMC_WSYNC(); reset_state(); } }
    return (badp || badv) && !(o.disableflags & DMC_DSBL_AUTORESET);
  }
  DMC_DEV bool bad_acc() {
    int bad = 0;
    FOR_LANES(i, L.d.nv) if (t_bad(S(qacc)[i])) bad = 1;
    return group_max<LPE>(bad) != 0;
  }

  // ---- pipeline -----------------------------------------------------------------------------
  // Position + velocity stage (mj_step1 without the checks).  partial = only what
  // the outputs need (the trailing mj_step1 of a legacy Physics.step()).
  DMC_DEV void stage_posvel(bool partial, int outmask, bool skipsensor, bool havekin = false) {
    if (!havekin) { kinematics(); DMC_PROF(PROF_KIN); DMC_TSUB(1, partial, 4); com_pos(); DMC_PROF(PROF_COM); DMC_TSUB(1, partial, 5); }
    if (!partial) { crb_mass_matrix(); DMC_PROF(PROF_CRB); DMC_TSUB(2, true, 4); }
    if (!partial || (outmask & (OUT_CONTACT | OUT_CONTACT_IDS))) { collision(); DMC_PROF(PROF_COLL); DMC_TSUB(2, !partial, 5); }
    if (!partial) { make_constraint(); DMC_PROF(PROF_CONSTR); DMC_TSUB(2, true, 6); }
    if (!skipsensor) sensors(DMC_STAGE_POS);
    DMC_PROF(PROF_SENS);
    DMC_TSUB(1, partial, 6);
    if (!havekin) { com_vel(); DMC_PROF(PROF_COMVEL); DMC_TSUB(1, partial, 7); }
    if (!partial) { passive_and_rne(); DMC_PROF(PROF_RNE); DMC_TSUB(2, true, 7); }
    if (!skipsensor) sensors(DMC_STAGE_VEL);
    DMC_PROF(PROF_SENS);
  }
  // Acceleration stage (mj_step2 without the integrator)
  DMC_DEV void stage_acc(bool disable_actuation, bool skipsensor) {
    fwd_actuation(disable_actuation); DMC_PROF(PROF_ACT); fwd_acceleration(); DMC_PROF(PROF_ACC); fwd_constraint();
    DMC_PROF(PROF_X8);      // (what fwd_constraint does after the solver's last marker: forces at the solution, J' f)
    if (!skipsensor) sensors_acc();
    DMC_PROF(PROF_SENS);
  }
  // investigation builds (-DDMC_TRACE_SUB=<region>): rows 4..7 of the wave trace are taken INSIDE one region of the
  // pipeline instead (the stage functions reach StepIO through its LDS copy, which follows StepOpts)
  DMC_DEV void sub_stamp(int row) {
#if defined(DMC_TRACE_SUB) && !defined(DMC_HOST_EMU)
    const StepIO<T>* iop = (const StepIO<T>*)((const unsigned char*)&o + (sizeof(StepOpts<T>) + 15) / 16 * 16);
    if (iop->trace && (threadIdx.x & 63) == 0) {
      const int epw = 64 / LPE, nitems = (iop->B + epw - 1) / epw;
      iop->trace[((size_t)(iop->trace_slot & 7) * 8 + row) * nitems + SI(imisc)[IM_ENV] / epw] = (int)(wall_clock64() & 0x7fffffffll);
    }
#else
    (void)row;
#endif
  }
  // wave trace: stage boundaries of the launch's first pass (rows 4..7 of the ring slot), one stamp per wave
  DMC_DEV void trace_stamp(const StepIO<T>& io, int env, int row) {
#if !defined(DMC_HOST_EMU) && !defined(DMC_TRACE_SUB)
    if (io.trace && (threadIdx.x & 63) == 0) {
      const int epw = 64 / LPE, nitems = (io.B + epw - 1) / epw;
      io.trace[((size_t)(io.trace_slot & 7) * 8 + row) * nitems + env / epw] = (int)(wall_clock64() & 0x7fffffffll);
    }
#else
    (void)io; (void)env; (void)row;
#endif
  }
  DMC_DEV void prof_begin() {
#if defined(DMC_PROFILE) && !defined(DMC_HOST_EMU)
    if (lane == 0) { for (int k = 0; k < 33; k++) SI(prof)[k] = 0; SI(prof)[33] = (int)__builtin_readcyclecounter(); }
#endif
  }
  DMC_DEV void prof_end(const StepIO<T>& io, int env) {
#if defined(DMC_PROFILE) && !defined(DMC_HOST_EMU)
    if (io.prof && lane == 0) for (int k = 0; k < 32; k++) io.prof[(size_t)k*io.B + env] += SI(prof)[k];
#else
    (void)io; (void)env;
#endif
  }
  // The three heavy stages are entered through out-of-line functions (StageFns):
  // each gets its own register allocation, so the peak pressure of one stage no
  // longer forces spills in the others, and there is one copy of the code.
#ifndef DMC_HOST_EMU
  DMC_DEV void call_posvel(bool partial, int outmask, bool skipsensor, bool havekin = false) {
    StageFns<T, LPE, LS>::posvel(ls, (const DMC_LDS StepOpts<T>*)&o, (DMC_LDS int*)mi, (DMC_LDS T*)mr, gc, (DMC_LDS T*)s,
                                 (DMC_LDS int*)si, lane, (partial ? 1 : 0) | (skipsensor ? 2 : 0) | (havekin ? 4 : 0), outmask);
  }
  DMC_DEV void call_acc(bool disable_actuation, bool skipsensor) {
    StageFns<T, LPE, LS>::acc(ls, (const DMC_LDS StepOpts<T>*)&o, (DMC_LDS int*)mi, (DMC_LDS T*)mr, gc, (DMC_LDS T*)s,
                              (DMC_LDS int*)si, lane, (disable_actuation ? 1 : 0) | (skipsensor ? 2 : 0));
  }
  DMC_DEV void call_euler() {
    StageFns<T, LPE, LS>::euler(ls, (const DMC_LDS StepOpts<T>*)&o, (DMC_LDS int*)mi, (DMC_LDS T*)mr, gc, (DMC_LDS T*)s,
                                (DMC_LDS int*)si, lane);
    time_ += o.timestep_d;
  }
#else
  DMC_DEV void call_posvel(bool partial, int outmask, bool skipsensor, bool havekin = false) { stage_posvel(partial, outmask, skipsensor, havekin); }
  DMC_DEV void call_acc(bool disable_actuation, bool skipsensor) { stage_acc(disable_actuation, skipsensor); }
  DMC_DEV void call_euler() { euler(); }
#endif
  DMC_DEV void load_ctrl_seq(const StepIO<T>& io, int env, int t) {
    env = late(env);
    const int B = io.B, nu = L.d.nu;
    DMC_WSYNC();
    FOR_LANES(i, nu) S(ctrl)[i] = io.ctrl_seq[((size_t)t*nu + i)*B + env];
    DMC_WSYNC();
  }
  DMC_DEV void store_seq(const StepIO<T>& io, int env, int t) {
    env = late(env);
    const int B = io.B;
    if (io.qpos_seq) FOR_LANES(i, L.d.nq) io.qpos_seq[((size_t)t*L.d.nq + i)*B + env] = S(qpos)[i];
    if (io.qvel_seq) FOR_LANES(i, L.d.nv) io.qvel_seq[((size_t)t*L.d.nv + i)*B + env] = S(qvel)[i];
    if (io.sensor_seq) FOR_LANES(i, L.d.nsensordata) io.sensor_seq[((size_t)t*L.d.nsensordata + i)*B + env] = S(sensordata)[i];
  }
  // mode: 0 = Physics.step(nstep) ; 1 = mj_forward ; 2 = mj_forward with actuation disabled ;
  //       3 = rollout: nstep env-steps of `nsub` substeps each, controls read from and
  //           per-step results written to (T, rows, B) sequence buffers -- legacy ordering
  //           (mj_step2 ... mj_step1) with NO redundant pass: the mj_step1 that closes
  //           env-step t is the position/velocity stage that opens env-step t+1.
  // With a stash (io.stash_r != null) a legacy Physics.step() does exactly the reference's work split
  // (engine.py:147-162): mj_step2 on the position / velocity stage the previous call's mj_step1 left behind
  // (reloaded from HBM instead of recomputed), ..., then a FULL mj_step1 whose results are stashed for the next
  // call.  Without one the opening stage is recomputed and the trailing mj_step1 only evaluates the outputs.
  // mode 4 / 5 = mj_step1 / mj_step2 as separate entry points (engine.py:156-162 calls them one after the other with
  // Python in between): mj_step1 leaves its stage in the stash, mj_step2 picks it up (and recomputes it when the
  // state was edited in between, where MuJoCo would integrate on stale derived arrays).
  DMC_DEV void run_split(const StepIO<T>& io, int env, int mode, int outmask, const Entry& en) {
    const bool stash = io.stash_r != nullptr;
    bool have = false;
    if (lane == 0) set_epoch(en.epoch);      // (kept in LDS, not in a register, for the whole launch)
    DMC_WSYNC();
    if (mode == 5 && stash) have = load_stash(io, env);
    load_state(io, env, have, en);
    if (mode == 4) {
      check_pos_vel();
      call_posvel(false, outmask, false);
      store_outputs(io, env, outmask);
      if (stash) store_stash(io, env, true);
      store_state(io, env);
      return;
    }
    int retried = 0;
    for (;;) {
      if (!have || retried) call_posvel(false, outmask, true);
      call_acc(false, false);
      if (!retried && bad_acc()) {
        if (lane == 0) SI(imisc)[IM_WARN + DMC_WARN_BADQACC]++;     // mj_checkAcc: reset + forward
        if (!(o.disableflags & DMC_DSBL_AUTORESET)) { DMC_WSYNC(); reset_state(); retried = 1; continue; }
      }
      break;
    }
    // acceleration-stage results; the position-stage arrays keep the values mj_step1 wrote
    store_outputs(io, env, outmask & (OUT_SENSOR | OUT_QACC | OUT_ACTUATOR | OUT_QFRC | OUT_CONTACT));
    call_euler();
    if (stash) store_stash(io, env, false);
    store_state(io, env);
  }
  // substep probe (StepIO::probe): slots [first, first + count) take the probed geom's current world position
  DMC_DEV TaskLds<T> task_lds() const {
    TaskLds<T> v;
    v.qpos = S(qpos); v.qvel = S(qvel); v.ctrl = S(ctrl); v.act = S(act); v.sensordata = S(sensordata); v.xpos = S(xpos);
    v.xmat = S(xmat); v.xipos = S(xipos); v.subtree_com = S(subtree_com); v.geom_xpos = S(geom_xpos); v.cvel = S(cvel);
    return v;
  }
  DMC_DEV void probe_store(const StepIO<T>& io, int env, int first, int count) {
    if constexpr (!kFeat) return;
    env = late(env);      // (addresses derived from env are formed here, not kept alive across the pass loop)
    T* p = io.probe;
    if (!p) return;
    const int cap = io.probe_cap, g = io.probe_geom;
    for (int t = first; t < first + count && t < cap; t++) FOR_LANES(k, 3) p[((size_t)t*3 + k)*io.B + env] = S(geom_xpos)[3*g + k];
  }
  DMC_DEV void run(const StepIO<T>& io, int env, int nstep, int legacy, int mode, int outmask, int nsub) {
    Entry en; EntryRegs er;
    entry_issue(L, o, io, env, lane, mode, legacy, &en, &er);
    entry_commit(L, io, env, lane, &en, er, s);
    run(io, env, nstep, legacy, mode, outmask, nsub, en);
  }
  // en: what entry_issue / entry_commit left for this env and the launch's (mode, legacy)
  // piece / npieces (sliced items of a queued launch, StepIO::slices): this call runs the physics steps
  // [piece * nstep / npieces, (piece + 1) * nstep / npieces) of a mode-0 launch from the state the previous piece stored;
  // the last piece also runs what ends the launch (the trailing mj_step1, outputs, the kinematic stash).  Every pass
  // rebuilds its derived arrays from (qpos, qvel, act, qacc_warmstart, time), so a cut between the integration of one
  // physics step and the position stage of the next recomputes nothing.
  DMC_DEV void run(const StepIO<T>& io, int env, int nstep, int legacy, int mode, int outmask, int nsub, const Entry& en,
                   int piece = 0, int npieces = 1) {
    const int launch_mode = mode;
    if (lane == 0) set_epoch(en.epoch);      // (kept in LDS, not in a register, for the whole launch)
    if (io.env_mode) {
      const int em = en.em;
      if (em == 2) return;
      if (em == 1 && (mode == 0 || mode >= 4)) mode = 2;
    }
    if constexpr (!kSlices) { piece = 0; npieces = 1; }
    if (mode != 0 && piece > 0) return;      // (an env the launch override turned into mj_forward: one pass, in the first piece)
    if (mode != 0) npieces = 1;
    if (mode >= 4) { run_split(io, env, mode, outmask, en); return; }
    prof_begin();
    const bool stash = io.stash_r != nullptr;
    bool have = false;
    if (stash && mode == 0 && legacy && o.integrator != DMC_INT_RK4) have = load_stash(io, env);
    load_state(io, env, have, en, kSlices && piece > 0);
    bool havekin = false;
    // (the stash was only prefetched for a launch that steps: an env switched to mj_forward by env_mode has none)
    if (piece == 0 && io.kstash && mode == 0 && launch_mode == 0 && legacy && !have && o.integrator != DMC_INT_RK4) {
      if (!en.fast) havekin = load_kstash(io, env);
      else if (en.kvalid) { havekin = true; DMC_WSYNC(); kstash_env_geoms(); }
    }
    DMC_PROF(PROF_LOAD);
    const bool stepping = mode == 0 || mode == 3;
    const int ntotal = mode == 3 ? nstep*nsub : nstep;
    // passes through the single stage_posvel()/stage_acc() call sites: ntotal full
    // passes (+ the trailing mj_step1 pass for legacy_step), or one pass for mj_forward
    const int npass = !stepping ? 1 : ntotal + ((legacy || mode == 3) ? 1 : 0);
    // legacy == 2: a legacy step whose trailing mj_step1 is followed by the rest of mj_forward at the new state (the
    // acceleration stage with its sensors, no integration) -- what the reference's composer observes after a control
    // step (mjcf/physics.py:341-342: the first observable read through a binding forwards the dirty physics)
    const bool fwd_after = kFeat && legacy == 2 && mode == 0;
    const bool last_piece = piece == npieces - 1;
    const int it_lo = npieces > 1 ? (int)((long long)piece * ntotal / npieces) : 0;
    const int it_hi = (npieces > 1 && !last_piece) ? (int)((long long)(piece + 1) * ntotal / npieces) : npass;
    for (int it = it_lo; it < it_hi; it++) {
      const bool trailing = stepping && it == ntotal;
      const bool partial = trailing && !(stash && mode == 0) && !fwd_after;
      if (mode == 3 && !trailing && it % nsub == 0 && io.ctrl_seq) load_ctrl_seq(io, env, it / nsub);
      if (stepping) { if (check_pos_vel()) { have = false; havekin = false; } }
      const int nstage = (stepping && !trailing && o.integrator == DMC_INT_RK4) ? 4 : 1;
      // Sensor values are overwritten by every step, so only the passes whose sensordata can be
      // read afterwards evaluate them: position/velocity sensors in the last pass of a launch and
      // at env-step boundaries of a rollout, acceleration sensors in the pass before those.
      const bool sens_pv = !stepping || it == npass - 1 || (mode == 3 && it % nsub == 0);
      const bool sens_acc = !stepping || (it == ntotal - 1 && !fwd_after) || (trailing && fwd_after) || (mode == 3 && (it + 1) % nsub == 0);
      int stage = 0, retried = 0;
      while (stage < nstage) {
        if (!(have && it == 0 && !retried)) call_posvel(partial, outmask, stage > 0 || !sens_pv, havekin && it == 0 && !retried && !trailing);
        if (it == 0 && stage == 0) trace_stamp(io, env, 4); else if (trailing) trace_stamp(io, env, 7);
        if (mode == 3 && stage == 0 && it > 0 && it % nsub == 0) store_seq(io, env, it / nsub - 1);
        if (mode == 0 && stage == 0 && it >= 1) probe_store(io, env, it - 1, 1);      // the state after `it` physics steps
        if (trailing && !fwd_after) break;
        call_acc(mode == 2, stage > 0 || !sens_acc);      // (the ONE call site of the acceleration stage)
        if (trailing) break;      // legacy_step 2: the launch ends with mj_forward's acceleration stage at the new state
        if (it == 0 && stage == 0) trace_stamp(io, env, 5);
        if (stage == 0 && stepping && !retried && bad_acc()) {
          if (lane == 0) SI(imisc)[IM_WARN + DMC_WARN_BADQACC]++;     // mj_checkAcc: reset + forward
          if (!(o.disableflags & DMC_DSBL_AUTORESET)) { DMC_WSYNC(); reset_state(); retried = 1; continue; }
        }
        if (stage == 0 && stepping && it == ntotal - 1) { dump_debug(io, env); if (mode == 0 && !legacy) store_outputs(io, env, outmask); }
        if (nstage > 1) rk4_stage(stage);
        stage++;
      }
      if (!stepping || trailing) break;
      if (nstage > 1) rk4_finish(); else call_euler();
      if (it == 0) trace_stamp(io, env, 6);
      DMC_PROF(PROF_EULER);
    }
    if constexpr (kSlices) if (!last_piece) { store_handoff(io, env); DMC_PROF(PROF_STORE); prof_end(io, env); return; }      // (the next piece takes it from here)
    if (!stepping) dump_debug(io, env);
    // an environment the launch override turned into mj_forward (just re-initialised) reports its one state in every slot
    if (!stepping && launch_mode == 0) probe_store(io, env, 0, nstep);
    if (!stepping || legacy || mode == 3) { DMC_PROF(PROF_TRAIL); store_outputs(io, env, outmask); }
    // the stash holds a complete position / velocity stage at the CURRENT state only after a legacy step (after an
    // mj_forward the Cholesky buffer holds the factor of H, not of M; a non-legacy mj_step ends before mj_step1)
    if (stash) store_stash(io, env, mode == 0 && legacy && !fwd_after);      // (after the forward the Cholesky buffer holds H's factor)
    // the trailing stage (partial or full) evaluated kinematics / COM frame / velocities at the state being stored
    if (io.kstash && mode == 0 && legacy && o.integrator != DMC_INT_RK4) store_kstash(io, env);
    store_state(io, env);
    DMC_PROF(PROF_STORE);
    prof_end(io, env);
  }
#undef MI
#undef MR
#undef GC
#undef S
#undef SG
#undef SI
#undef FOR_LANES
};


#ifndef DMC_HOST_EMU
template <typename T, int LPE, typename LS>
struct StageFns {
  typedef StepCore<T, LPE, LS> Core;
  static DMC_FN void posvel(LS ls, const DMC_LDS StepOpts<T>* o, DMC_LDS int* mi, DMC_LDS T* mr, const int* gc, DMC_LDS T* s,
                            DMC_LDS int* si, int lane, int flags, int outmask) {
    Core c(ls, *(const StepOpts<T>*)o, (const int*)mi, (const T*)mr, gc, (T*)s, (int*)si, lane);
    c.stage_posvel(flags & 1, outmask, flags & 2, flags & 4);
  }
  static DMC_FN void acc(LS ls, const DMC_LDS StepOpts<T>* o, DMC_LDS int* mi, DMC_LDS T* mr, const int* gc, DMC_LDS T* s,
                         DMC_LDS int* si, int lane, int flags) {
    Core c(ls, *(const StepOpts<T>*)o, (const int*)mi, (const T*)mr, gc, (T*)s, (int*)si, lane);
    c.stage_acc(flags & 1, flags & 2);
  }
  static DMC_FN void ns_build(LS ls, const DMC_LDS StepOpts<T>* o, DMC_LDS int* mi, DMC_LDS T* mr, const int* gc, DMC_LDS T* s,
                              DMC_LDS int* si, int lane, int nf) {
    if constexpr (LS::kNV > 0 && LS::kNV <= 32 && LPE == 64) {
      Core c(ls, *(const StepOpts<T>*)o, (const int*)mi, (const T*)mr, gc, (T*)s, (int*)si, lane);
      c.template noslip_build_A_lanes<LS::kNV>((const DMC_LDS T*)c.M_factor(), nf, c.row_map());
    }
  }
  static DMC_FN void euler(LS ls, const DMC_LDS StepOpts<T>* o, DMC_LDS int* mi, DMC_LDS T* mr, const int* gc, DMC_LDS T* s,
                           DMC_LDS int* si, int lane) {
    Core c(ls, *(const StepOpts<T>*)o, (const int*)mi, (const T*)mr, gc, (T*)s, (int*)si, lane);
    c.euler_state();
  }
};
#endif

}  // namespace dmc
