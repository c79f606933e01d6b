// step_kernel.hip.h -- the __global__ wrapper around StepCore and its launcher.
// Included by step_kernels_f32.hip (fp-contract=fast) and step_kernels_f64.hip
// (fp-contract=off, so the fp64 instantiation rounds like the oracle).
#pragma once
#include <hip/hip_runtime.h>

#ifndef DMC_STEP_CORE_HEADER   // tuning studies build variants of the core from other files
#define DMC_STEP_CORE_HEADER "step_core.h"
#endif
#include DMC_STEP_CORE_HEADER
#ifdef DMC_TASK_HEADER      // (a specialisation plugin with a task epilogue: dm_control_amd/suite/fused_env.py)
#define DMC_TASK_IN_KERNEL 1
#include DMC_TASK_HEADER
#endif

namespace dmc {

struct LaunchGeom {
  int lpe;             // lanes per environment: 64, 32 or 16
  int waves;           // wavefronts per workgroup (1..5: five where only a five-wave group holds one more environment)
  int envs_per_block;  // waves * 64 / lpe
  int lds_bytes;       // dynamic LDS per workgroup
  int grid;            // workgroups: one per envs_per_block environments, or (queue != 0) only the resident ones
  int queue;           // 1: the batch is larger than what is resident at once -- waves claim further items from StepIO::work
  int blocks_per_cu;   // workgroups of this shape resident on one CU (LDS- and VGPR-limited)
  int static_id;       // >= 0: layout equals the baked layout with this id (specialised kernel)
};

#ifndef DMC_MIN_WAVES
#define DMC_MIN_WAVES 2   // waves per SIMD the register allocator must leave room for
#endif
#ifndef DMC_MAX_THREADS
#define DMC_MAX_THREADS 320   // workgroups of up to five waves: five 62-dof environments share a CU's LDS (one table copy)
#endif

// Build-generated: LDS layouts of the suite models, baked in as compile-time
// constants (dm_control_amd/build.py -> gen_static_layouts).  A batch whose
// runtime layout equals a baked one runs the specialised instantiation, in which
// every LDS offset is an immediate and every size a constant.
#if defined(DMC_LAYOUTS_HEADER)      // a specialisation plugin (step_kernel_spec.hip): ONE model's layout, generated at run time
#include DMC_LAYOUTS_HEADER
#elif defined(DMC_PROFILE) && __has_include("static_layouts_prof.gen.h")
#include "static_layouts_prof.gen.h"
#elif __has_include("static_layouts.gen.h")
#include "static_layouts.gen.h"
#endif
#ifndef DMC_NSTATIC
#define DMC_NSTATIC 0
#endif

// bytes of the per-workgroup header in LDS that holds the scalar options (read by
// the out-of-line stage functions, which cannot see kernel arguments)
// ... followed by a copy of StepIO: its ~60 pointers do not fit the SGPR file next to everything else, and as kernel
// arguments they were spilled to VGPR lanes in the 16-dword tuples they were loaded in -- every use of ONE pointer
// read 16 lanes back (2 400 v_readlane in the cheetah kernel).  From LDS a use is one ds_read_b64.
template <typename T> constexpr int stepopts_lds_bytes() { return (int)((sizeof(StepOpts<T>) + 15) / 16 * 16); }
template <typename T> constexpr int opts_lds_bytes() { return stepopts_lds_bytes<T>() + (int)((sizeof(StepIO<T>) + 15) / 16 * 16); }

// QUEUE = false: every wave has exactly one item (the batch fits the resident grid) -- no claim loop, so nothing of
// the body is "loop invariant": with the loop the kernel arguments are hoisted out of it, spilled to VGPR lanes
// (106 SGPRs hold a fraction of StepIO's ~60 pointers) and read back with v_readlane where they are used.
template <typename T, int LPE, typename LS, bool QUEUE>
__device__ __forceinline__ void step_kernel_body(LS ls, const StepOpts<T>& o_arg, const int* __restrict__ g_mi,
                                                 const T* __restrict__ g_mr, const int* __restrict__ g_mc, const StepIO<T>& io_arg, int nstep, int legacy,
                                                 int mode, int outmask, int nsub) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const StepLayout& L = ls.get();
  StepOpts<T>* o_lds = reinterpret_cast<StepOpts<T>*>(smem);
  unsigned char* tables = smem + opts_lds_bytes<T>();
  int* mi = reinterpret_cast<int*>(tables);
  T* mr = reinterpret_cast<T*>(tables + (size_t)L.n_mi * sizeof(int));
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int t_entry = io_arg.trace ? (int)(wall_clock64() & 0x7fffffffll) : 0;
  typedef StepCore<T, LPE, LS> Core;
  constexpr int epw = 64 / LPE;
  // XCD-aware mapping: workgroup b runs on XCD b % 8, and each XCD has its own L2.
  // Give every XCD one contiguous range of environments, so that a 128-B line of
  // an SoA row (32 fp32 envs = several workgroups) is fetched into one L2 only.
  const int nblk = gridDim.x, xcd = blockIdx.x & 7, q = nblk >> 3, r = nblk & 7;
  const int lblk = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wpb = nthr >> 6;      // wave-uniform: the loop state lives in SGPRs
  const int nitems = (io_arg.B + epw - 1) / epw, nwaves = nblk * wpb;
  // The first item's state and kinematic stash are requested before the tables are staged and written to the env's
  // LDS scratch before the barrier: one HBM round trip for tables, state and stash.
  const size_t tables_bytes = (size_t)L.n_mi * sizeof(int) + (size_t)L.n_mr_lds * sizeof(T) + (L.d.coldlds ? (size_t)L.n_mc * sizeof(int) : 0);
  const size_t env_bytes = (size_t)L.n_sr * sizeof(T) + (size_t)L.n_si * sizeof(int);
  typename Core::Entry en;
  typename Core::EntryRegs er;
  // Queued launches (QUEUE: the grid is only the RESIDENT workgroups of a larger batch): one queue PER XCD.  Queue x holds
  // the items at positions x, x + NX, x + 2 NX ... of the hand-out order (longest first, dealt round: the queues' predicted
  // sums are balanced) as `npieces` rounds -- piece s of each of its items before piece s + 1 of any (StepIO::slices) --
  // and is served by the waves that RUN on XCD x (HW_REG_XCC_ID: the hardware's answer, not an assumption about
  // dispatch), so that everything an environment leaves in global memory between its pieces is written and read
  // through ONE L2.  (The per-XCD L2s are not coherent with each other: two of them holding dirty lines of the same
  // per-env scratch could write them back in either order.)  A position is claimed with one atomicAdd on the queue's head.
  // the argument structs go to LDS first (read from there by everything behind the barrier): with the queue's claim -- an
  // atomic round trip -- in front of these copies the compiler parked both structs in every lane's scratch (+ 500 B)
  if (tid == 0) *o_lds = o_arg;
  StepIO<T>* io_lds = reinterpret_cast<StepIO<T>*>(smem + stepopts_lds_bytes<T>());
  if (tid == 64 % nthr) *io_lds = io_arg;
  const StepIO<T>& io = *io_lds;
  int* const q_work = QUEUE ? io_arg.work : nullptr;
  int* const q_prog = QUEUE ? io_arg.prog : nullptr;
  const int q_slices = QUEUE ? io_arg.slices : 0, q_nxcd = QUEUE ? io_arg.nxcd : 1;
  const int NX = (QUEUE && q_work && q_nxcd > 1) ? q_nxcd : 1;
  const int xq = NX > 1 ? (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) % (unsigned)NX) : 0;      // HW_REG_XCC_ID[3:0]
  const int nq_items = (nitems - xq + NX - 1) / NX;      // items of this wave's queue
  const int npieces = (Core::kSlices && QUEUE && q_work && q_prog && q_slices > 1 && mode == 0) ? q_slices : 1;
  const int nslots = nq_items * npieces;
  int* const head = (QUEUE && q_work) ? q_work + 32 * (1 + xq) : nullptr;      // (a 128-byte line per head)
  int slot = lblk * wpb + wave, piece = 0;
  bool have_item = slot < nitems;
  if (QUEUE && head) {
    int nx = 0;
    if ((tid & 63) == 0) nx = atomicAdd(head, 1);
    nx = __builtin_amdgcn_readfirstlane(nx);
    have_item = nx < nslots;
    for (piece = 0; nx >= nq_items && piece < npieces - 1; piece++) nx -= nq_items;      // (at most seven trips: no division sequence)
    slot = nx * NX + xq;
  }
  int env0 = 0;
  if (have_item && piece == 0) {
    const int item = io_arg.order ? io_arg.order[slot] : slot;
    env0 = item * epw + ((tid / LPE) & (epw - 1));
    if (env0 >= io_arg.B) env0 = io_arg.B - 1;      // ragged last wave: loads the last environment, runs nothing
    Core::entry_issue(L, o_arg, io_arg, env0, tid % LPE, mode, legacy, &en, &er);
  } else if (have_item) {      // (a wave whose first claim is already a later piece: loads in place, as every later claim does)
    const int item = io_arg.order ? io_arg.order[slot] : slot;
    int e0 = item * epw + ((tid / LPE) & (epw - 1));
    if (e0 >= io_arg.B) e0 = io_arg.B - 1;
    en.em = io_arg.env_mode ? io_arg.env_mode[e0] : 0; en.fast = 0; en.kvalid = 0; en.epoch = *io_arg.epoch;
  }
  // stage the model constant tables once per workgroup (shared by all its envs)
  for (int i = tid; i < L.n_mi; i += nthr) mi[i] = g_mi[i];
  for (int i = tid; i < L.n_mr_lds; i += nthr) mr[i] = g_mr[i];   // large models: only the hot tables (StepLayout::n_mr_lds)
  // small models: the cold tables ride along in LDS (after the real tables); large ones read them from global memory
  int* mc_lds = reinterpret_cast<int*>(tables + (size_t)L.n_mi * sizeof(int) + (size_t)L.n_mr_lds * sizeof(T));
  const size_t cold_bytes = L.d.coldlds ? (size_t)L.n_mc * sizeof(int) : 0;
  if (L.d.coldlds) for (int i = tid; i < L.n_mc; i += nthr) mc_lds[i] = g_mc[i];
  if (have_item && piece == 0) Core::entry_commit(L, io_arg, env0, tid % LPE, &en, er, reinterpret_cast<T*>(tables + tables_bytes + (size_t)(tid / LPE) * env_bytes));
  __syncthreads();
  // An item is the 64 / LPE environments one wave steps together.  Without a queue every wave has the item of its
  // position in the grid.  With one, a wave that finishes claims the next position of its XCD's queue: the waves of a
  // workgroup do not wait for its slowest environment, and a launch is not a whole number of rounds (a fallen 62-dof
  // humanoid steps several times longer than a standing one; with 4-wave workgroups handed out whole, one launch per
  // env-step ran 1.5x longer than the rollout).  Sliced items (npieces > 1): a piece is `nstep / npieces` physics steps of
  // the item; piece s waits for piece s - 1 of its item, which was claimed earlier from the same head and is therefore
  // running or done -- no wait can be circular.  Round 6, config 4 on one box: whole items 8.90 ms per launch (the queue
  // runs dry at 62 % of the launch, the waves then idle behind items of up to 6 ms whose cost the previous launch
  // predicts with a rank correlation of 0.53), pieces 7.4 ms.
  if (have_item) for (;;) {
    // The per-lane pointers are re-derived from the thread index on every trip (behind an opaque copy, so that the
    // compiler does not hoist them): otherwise they stay live across the out-of-line stage calls of run() and are
    // spilled to scratch memory there -- 11 VGPRs on the cheetah kernel, 6 x the algorithmic HBM writes of a launch.
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int g = t / LPE, lane = t % LPE;
    unsigned char* base = tables + (size_t)L.n_mi * sizeof(int) + (size_t)L.n_mr_lds * sizeof(T) + cold_bytes + (size_t)g * env_bytes;
    T* s = reinterpret_cast<T*>(base);
    int* si = reinterpret_cast<int*>(base + (size_t)L.n_sr * sizeof(T));
    StepCore<T, LPE, LS> core(ls, *o_lds, mi, mr, L.d.coldlds ? (const int*)mc_lds : g_mc, s, si, lane);
    // longest first: an environment that took long last time (a fallen humanoid with 20 contacts) is started early,
    // so that the launch does not end waiting for one that was started last
    const int item = io.order ? io.order[slot] : slot;
    const int env = item * epw + (g & (epw - 1));
    int* tr = io.trace ? io.trace + (size_t)(io.trace_slot & 7) * 8 * nitems : nullptr;
    const long long tw0 = tr ? (long long)wall_clock64() : 0;
    if (QUEUE && piece > 0)      // (the state itself is read with loads that bypass this CU's L1: StepCore::load_handoff)
      while (__hip_atomic_load(q_prog + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < piece) __builtin_amdgcn_s_sleep(8);
    const long long tw1 = tr ? (long long)wall_clock64() : 0;
    const long long t0 = io.cost ? (long long)__builtin_readcyclecounter() : 0;
    if (tr && (threadIdx.x & 63) == 0 && piece == 0) { tr[item] = t_entry; tr[nitems + item] = (int)(wall_clock64() & 0x7fffffffll); }
    if (env < io.B) core.run(io, env, nstep, legacy, mode, outmask, nsub, en, piece, npieces);
#ifdef DMC_TASK_HEADER
    // the generated task layer of this model's task (suite/fused_env.py), by one lane per environment: it reads what the
    // launch has just stored (loads that bypass the L1, after the wave's stores are acknowledged)
    if (io.task_args && mode == 0 && piece == npieces - 1 && env < io.B) {
      // (what the function reads from global memory -- and the state a restart overwrites -- was stored by this wave: drained first)
      if (dmc_task::kReadsGlobal) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) dmc_task::task_post(*(const dmc_task::PostArgs*)io.task_args, env, core.task_lds());
    }
#endif
    if (tr && (threadIdx.x & 63) == 0 && piece == npieces - 1) {
      tr[2*nitems + item] = (int)(wall_clock64() & 0x7fffffffll);
      tr[3*nitems + item] = (int)blockIdx.x;
    }
    if (tr && (threadIdx.x & 63) == 0) {      // rows 4 / 5: ticks this item's pieces waited for their predecessors / ran (scripts/queue_wait_probe.py)
      const int wt = (int)(tw1 - tw0), rn = (int)((long long)wall_clock64() - tw1);
      if (piece == 0) { __hip_atomic_store(tr + 4*nitems + item, wt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(tr + 5*nitems + item, rn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else { atomicAdd(tr + 4*nitems + item, wt); atomicAdd(tr + 5*nitems + item, rn); }
    }
    if (io.cost && (threadIdx.x & 63) == 0) {
      long long dt = ((long long)__builtin_readcyclecounter() - t0) >> 6;
      dt = dt < 1 ? 1 : (dt > 0x07ffffff ? 0x07ffffff : dt);
      // (the pieces of an item may run on different CUs: the sum is formed where atomics execute, not in an L1)
      if (piece == 0) __hip_atomic_store(io.cost + item, (int)dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else atomicAdd(io.cost + item, (int)dt);
    }
    if (QUEUE && piece < npieces - 1) {
      // the hand-off record was written with write-through (sc1) stores: once they are acknowledged the flag may go
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(q_prog + item, piece + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!QUEUE || !head) break;
    int nx = 0;
    if ((threadIdx.x & 63) == 0) nx = atomicAdd(head, 1);
    nx = __builtin_amdgcn_readfirstlane(nx);
    if (nx >= nslots) break;
    for (piece = 0; nx >= nq_items && piece < npieces - 1; piece++) nx -= nq_items;
    slot = nx * NX + xq;
    const int nitem = io.order ? io.order[slot] : slot;
    int nenv = nitem * epw + ((threadIdx.x / LPE) & (epw - 1));
    if (nenv >= io.B) nenv = io.B - 1;
    // (later items load in place, inside run(): up here the registers are full of what the loop keeps alive)
    en.em = io.env_mode ? io.env_mode[nenv] : 0; en.fast = 0; en.kvalid = 0;
  }
  if (QUEUE && q_work && (threadIdx.x & 63) == 0) {
    // every wave makes exactly one failing claim before it gets here, so the last wave to arrive can re-arm the queues
    // for the next launch on the stream
    if (atomicAdd(q_work + 1, 1) == nwaves - 1) {
      atomicExch(q_work + 1, 0);
      for (int x = 0; x < NX; x++) atomicExch(q_work + 32 * (1 + x), 0);
      if (npieces > 1) for (int i = 0; i < nitems; i++) q_prog[i] = 0;      // (every piece is complete: nobody reads it any more)
    }
  }
}

template <typename T, int LPE, bool QUEUE>
__global__ void __launch_bounds__(DMC_MAX_THREADS, DMC_MIN_WAVES)
step_kernel(const StepLayout* __restrict__ Lp, StepOpts<T> o, const int* __restrict__ g_mi, const T* __restrict__ g_mr,
            const int* __restrict__ g_mc, StepIO<T> io, int nstep, int legacy, int mode, int outmask, int nsub) {
  // generic kernel: the layout lives in device memory (uniform scalar loads); taking
  // the address of a by-value kernel argument would copy it to scratch
  DynLayoutSrc ls; ls.p = Lp;
  step_kernel_body<T, LPE, DynLayoutSrc, QUEUE>(ls, o, g_mi, g_mr, g_mc, io, nstep, legacy, mode, outmask, nsub);
}

#if DMC_NSTATIC > 0
template <int SID> struct StaticLayout;
#define DMC_DEF_STATIC(ID)                                                            \
  static __device__ const StepLayout kStaticLayout##ID = DMC_STATIC_LAYOUT_##ID;     \
  template <> struct StaticLayout<ID> {                                               \
    static constexpr int kNV = DMC_STATIC_NV_##ID;   /* compile-time nv: register-resident Cholesky */ \
    static constexpr StepLayout kL = DMC_STATIC_LAYOUT_##ID;                                      \
    static constexpr int kJGlobal = kL.d.jglobal;   /* (a small batch keeps the contact rows in LDS: jlevel of step_tables_build) */ \
    static constexpr int kNKin = kL.s_qM - kL.s_xpos;   /* reals of the kinematic stash */       \
    static constexpr int kTreeMax = kL.d.treemax;   /* > 0: factorisations run the kinematic trees side by side */ \
    static constexpr int kTreeUni = kL.d.treeuni;   /* trees of equal size: the block structure of M's factor is a compile-time fact */ \
    __device__ __forceinline__ const StepLayout& get() const { return kStaticLayout##ID; } \
  };
DMC_STATIC_IDS(DMC_DEF_STATIC)
#undef DMC_DEF_STATIC

template <typename T, int LPE, int SID, bool QUEUE>
__global__ void __launch_bounds__(DMC_MAX_THREADS, DMC_MIN_WAVES)
step_kernel_static(StepOpts<T> o, const int* __restrict__ g_mi, const T* __restrict__ g_mr,
                   const int* __restrict__ g_mc, StepIO<T> io, int nstep, int legacy, int mode, int outmask, int nsub) {
  step_kernel_body<T, LPE, StaticLayout<SID>, QUEUE>(StaticLayout<SID>(), o, g_mi, g_mr, g_mc, io, nstep, legacy, mode, outmask, nsub);
}
#endif

// (internal linkage: the fp32 units instantiate launch_step_t<float> with DIFFERENT instance lists -- as an inline
// function with external linkage the linker would keep one body for both, and the large models would silently run the
// generic kernel: 3x slower, same results)
template <typename T>
static hipError_t launch_step_t(const LaunchGeom& g, hipStream_t stream, const StepLayout* d_layout, const StepOpts<T>& o,
                                const int* g_mi, const T* g_mr, const int* g_mc, const StepIO<T>& io, int nstep, int legacy, int mode, int outmask, int nsub) {
  const dim3 grid(g.grid), block(g.waves * 64);
#define DMC_LAUNCH_Q(LPE, Q)                                                                                    \
  {                                                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<T, LPE, Q>),                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);                \
    if (e != hipSuccess) return e;                                                                              \
    hipLaunchKernelGGL((step_kernel<T, LPE, Q>), grid, block, g.lds_bytes, stream, d_layout, o, g_mi, g_mr, g_mc, io, nstep, \
                       legacy, mode, outmask, nsub);                                                                  \
  }
#define DMC_LAUNCH(LPE) { if (io.work) DMC_LAUNCH_Q(LPE, true) else DMC_LAUNCH_Q(LPE, false) }
#define DMC_LAUNCH_STATIC_Q(LPE, SID, Q)                                                                        \
  {                                                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel_static<T, LPE, SID, Q>),      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);                \
    if (e != hipSuccess) return e;                                                                              \
    hipLaunchKernelGGL((step_kernel_static<T, LPE, SID, Q>), grid, block, g.lds_bytes, stream, o, g_mi, g_mr, g_mc, io,  \
                       nstep, legacy, mode, outmask, nsub);                                                           \
    return hipGetLastError();                                                                                   \
  }
#define DMC_LAUNCH_STATIC(LPE, SID) { if (io.work) DMC_LAUNCH_STATIC_Q(LPE, SID, true) else DMC_LAUNCH_STATIC_Q(LPE, SID, false) }
#if DMC_NSTATIC > 0
  // specialised instantiations exist for (static id, lanes) pairs listed in DMC_STATIC_INSTANCES; the fp32 ones are
  // split over two compilation units (build.py: the large models are scheduled with a different strategy)
#if defined(DMC_UNIT_ILP)
#define DMC_UNIT_INSTANCES(X) DMC_STATIC_INSTANCES_ILP(X)
#elif defined(DMC_UNIT_STD)
#define DMC_UNIT_INSTANCES(X) DMC_STATIC_INSTANCES_STD(X)
#else
#define DMC_UNIT_INSTANCES(X) DMC_STATIC_INSTANCES(X)
#endif
  // (a specialised kernel built without the optional launch features -- step_core.h kFeat -- hands such launches to the generic one)
  const bool need_feat = legacy == 2 || io.probe != nullptr || o.integrator == DMC_INT_IMPLICITFAST;
#define DMC_X(SID, LPE) if (g.static_id == SID && g.lpe == LPE && (DMC_STATIC_FEATURES || !need_feat)) DMC_LAUNCH_STATIC(LPE, SID)
  DMC_UNIT_INSTANCES(DMC_X)
#undef DMC_X
#endif
#if defined(DMC_UNIT_ILP)
  return hipErrorInvalidValue;      // (this unit holds no generic kernel)
#else
  if (g.lpe == 64) DMC_LAUNCH(64)
  else if (g.lpe == 32) DMC_LAUNCH(32)
  else if (g.lpe == 16) DMC_LAUNCH(16)
  else return hipErrorInvalidValue;
#endif
#undef DMC_LAUNCH
#undef DMC_LAUNCH_STATIC
#undef DMC_LAUNCH_Q
#undef DMC_LAUNCH_STATIC_Q
  return hipGetLastError();
}

}  // namespace dmc
