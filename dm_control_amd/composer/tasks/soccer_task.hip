// The task layer of locomotion.soccer 2-vs-2 (BASELINE config 5) around the physics launch of a control step, as TWO
// kernels instead of ~150 small tensor operations (composer/tasks/soccer.py is the tensor form, kept as the CPU tier's
// path and as the checker of this one: tests/test_gpu_composer.py::test_soccer_task_kernels_equal_the_tensor_task_layer).
//
//   soccer_pre   composer.Environment.step up to the launch (composer/environment.py:412-428 of the reference): for the
//                environments whose episode ended -- mj_resetData, UniformInitializer (soccer/initializers.py:96-127),
//                detector / prev_action reset, launch override `mj_forward without actuation`; for all -- the
//                players' actions into ctrl (task.py:211-213), the throw-in (task.py:128-135, :215-217), the
//                retained detections cleared (position_detector.py before_step)
//   soccer_post  after the launch: the three PositionDetectors on the substep probe's trace with the per-substep
//                semantics (position_detector.py:200-260, retain_substep_detections), rewards / discount / termination
//                (task.py:160-209), mjWARN_* -> episode over, time limit, step types, the reset flags, and the part of
//                the observation that is not a row of mjData: prev_action, goal / field corners in the player's frame
//                and the stats_* scalars (soccer/observables.py:262-375) -- the rest is one launch of the gather kernel
//
// One thread per environment (pre) / per (player, environment) (post); every array is (rows, B) with the environment
// index fastest, so neighbouring threads read neighbouring addresses.  DMC_SOCCER_REAL is float or double (prepended
// by soccer.py).  Contraction is off: the arithmetic rounds like the tensor form's separate operations.
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)
typedef DMC_SOCCER_REAL T;

struct SoccerArgs {
  int B, nq, nv, nu, nsub, rounds;
  int ball_q, ball_v, ball_geom, ball_linvel;
  int place_rows[15];      // qpos rows: ball x y z, the players' (root_x, root_y) x 4, the players' steer x 4
  int ctrl_rows[12];       // ctrl rows of (roll, steer, kick) x 4
  int root[4];             // body ids of the players' heads
  int ncomp;               // width of a player's computed block
  double time_limit;
  T spawn_ratio, ball_z, pi;
  T* qpos; T* qvel; T* ctrl; T* warm; double* time; int* env_mode; int* warning;
  const T* qpos0; const T* size;
  unsigned char* state;    // (3, B): home goal, away goal, field (inverted) detections
  const T* lo; const T* hi;      // (3, 3, B) detector bounds
  T* prev_action;          // (4, 3, B)
  const T* geom_xpos; const T* xpos; const T* xmat; const T* cvel; const T* sensordata;
  const T* action;         // (B, 12)
  const T* u_place;        // (rounds, 5, 3, B) uniform [0, 1)
  const T* u_shrink;       // (2, B) uniform [0, 1)
  const T* trace;          // (capacity, 3, B) ball position after every physics step of the launch
  unsigned char* reset_next;
  T* reward; T* discount; int* step_type; T* comp;
};

extern "C" __global__ void __launch_bounds__(256) soccer_pre_kernel(SoccerArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int B = a.B;
  if (e >= B) return;
  const bool first = a.reset_next[e] != 0;
  if (first) {
    for (int i = 0; i < a.nq; i++) a.qpos[(size_t)i * B + e] = a.qpos0[i];
    for (int i = 0; i < a.nv; i++) { a.qvel[(size_t)i * B + e] = 0; a.warm[(size_t)i * B + e] = 0; }
    a.time[e] = 0;
    // UniformInitializer: all candidate placements are drawn, the first one without two entities closer than 1.5 is taken
    const T sx = a.size[e] * a.spawn_ratio, sy = a.size[B + e] * a.spawn_ratio;
    int take = a.rounds - 1;
    for (int r = a.rounds - 2; r >= 0; r--) {
      T x[5], y[5];
      for (int n = 0; n < 5; n++) {
        x[n] = (a.u_place[((size_t)(r * 5 + n) * 3 + 0) * B + e] * 2 - 1) * sx;
        y[n] = (a.u_place[((size_t)(r * 5 + n) * 3 + 1) * B + e] * 2 - 1) * sy;
      }
      bool close = false;
      for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) if (i != j) {
        const T dx = x[i] - x[j], dy = y[i] - y[j];
        close |= sqrt(dx * dx + dy * dy) < (T)1.5;
      }
      if (!close) take = r;
    }
    T v[15];
    for (int n = 0; n < 5; n++) {
      const T ux = a.u_place[((size_t)(take * 5 + n) * 3 + 0) * B + e] * 2 - 1;
      const T uy = a.u_place[((size_t)(take * 5 + n) * 3 + 1) * B + e] * 2 - 1;
      const T uz = a.u_place[((size_t)(take * 5 + n) * 3 + 2) * B + e] * 2 - 1;
      if (n == 0) { v[0] = ux * sx; v[1] = uy * sy; v[2] = a.ball_z; }
      else { v[3 + 2 * (n - 1)] = ux * sx; v[4 + 2 * (n - 1)] = uy * sy; v[11 + (n - 1)] = uz * a.pi; }
    }
    for (int k = 0; k < 15; k++) a.qpos[(size_t)a.place_rows[k] * B + e] = v[k];
    for (int d = 0; d < 3; d++) a.state[d * B + e] = 0;
  }
  a.env_mode[e] = first ? 1 : 0;
  // before_step of the task: actions (an environment that restarts keeps mj_resetData's zero controls), prev_action
  for (int i = 0; i < a.nu; i++) if (first) a.ctrl[(size_t)i * B + e] = 0;
  for (int k = 0; k < 12; k++) {
    const T u = a.action[(size_t)e * 12 + k];
    a.prev_action[(size_t)k * B + e] = u;
    if (!first) a.ctrl[(size_t)a.ctrl_rows[k] * B + e] = u;
  }
  // throw-in: the ball left the court in the previous control step
  if (a.state[2 * B + e]) {
    const T lo = (T)0.7, hi = (T)0.9, w = hi - lo;
    for (int c = 0; c < 2; c++) {
      const T shrink = lo + w * a.u_shrink[c * B + e];
      a.qpos[(size_t)(a.ball_q + c) * B + e] = a.geom_xpos[(size_t)(3 * a.ball_geom + c) * B + e] * shrink;
    }
    a.qpos[(size_t)(a.ball_q + 2) * B + e] = a.ball_z;
    for (int i = 0; i < 6; i++) a.qvel[(size_t)(a.ball_v + i) * B + e] = 0;
  }
  // before_step of the detectors: retained detections (the goals') last one control step
  a.state[e] = 0; a.state[B + e] = 0;
}

__device__ inline T norm2(T x, T y) { return sqrt(x * x + y * y); }
__device__ inline T norm3(T x, T y, T z) { return sqrt(x * x + y * y + z * z); }

extern "C" __global__ void __launch_bounds__(256) soccer_post_kernel(SoccerArgs a) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int B = a.B;
  if (tid >= 4 * B) return;
  const int e = tid % B, k = tid / B;
  // detectors: every thread of the environment evaluates them (15 loads), player 0's stores them
  bool st[3];
  for (int d = 0; d < 3; d++) {
    const T l0 = a.lo[(size_t)(d * 3 + 0) * B + e], l1 = a.lo[(size_t)(d * 3 + 1) * B + e], l2 = a.lo[(size_t)(d * 3 + 2) * B + e];
    const T h0 = a.hi[(size_t)(d * 3 + 0) * B + e], h1 = a.hi[(size_t)(d * 3 + 1) * B + e], h2 = a.hi[(size_t)(d * 3 + 2) * B + e];
    bool any = false, last = false;
    for (int n = 0; n < a.nsub; n++) {
      const T p0 = a.trace[(size_t)(n * 3 + 0) * B + e], p1 = a.trace[(size_t)(n * 3 + 1) * B + e], p2 = a.trace[(size_t)(n * 3 + 2) * B + e];
      const bool inside = p0 > l0 && p0 < h0 && p1 > l1 && p1 < h1 && p2 > l2 && p2 < h2;
      last = d == 2 ? !inside : inside;      // the field detector is inverted
      any |= last;
    }
    st[d] = d == 2 ? last : (a.state[d * B + e] != 0) | any;
  }
  const bool first = a.env_mode[e] != 0;
  // arena.detected_goal (pitch.py:574-580): the home goal is looked at first -- the ball in it means AWAY scored
  const bool away_scored = st[0], home_scored = st[1] && !st[0];
  if (k == 0) {
    for (int d = 0; d < 3; d++) a.state[d * B + e] = st[d] ? 1 : 0;
    int w = 0;
    for (int i = 0; i < 8; i++) { w += a.warning[(size_t)i * B + e]; a.warning[(size_t)i * B + e] = 0; }
    const bool diverged = w > 0;
    const bool goal = st[0] || st[1];
    const bool terminating = goal || a.time[e] >= a.time_limit - 1e-9 || diverged;
    T r = (T)(home_scored ? 1 : 0) - (T)(away_scored ? 1 : 0);
    if (diverged || first) r = 0;
    a.reward[e] = r; a.reward[B + e] = r; a.reward[2 * B + e] = -r; a.reward[3 * B + e] = -r;
    a.discount[e] = first ? (T)1 : ((goal || diverged) ? (T)0 : (T)1);
    a.step_type[e] = first ? 0 : (terminating ? 2 : 1);
    a.reset_next[e] = (terminating && !first) ? 1 : 0;
  }
  // what player k observes beyond the rows of mjData
  const bool away = k >= 2;
  const int mate = k ^ 1;
  const int b = a.root[k], bm = a.root[mate];
  T pos[3], pm[3], R[9], ball[3], blin[3];
  for (int i = 0; i < 3; i++) {
    pos[i] = a.xpos[(size_t)(3 * b + i) * B + e]; pm[i] = a.xpos[(size_t)(3 * bm + i) * B + e];
    ball[i] = a.geom_xpos[(size_t)(3 * a.ball_geom + i) * B + e]; blin[i] = a.sensordata[(size_t)(a.ball_linvel + i) * B + e];
  }
  for (int i = 0; i < 9; i++) R[i] = a.xmat[(size_t)(9 * b + i) * B + e];
  const T cvx = a.cvel[(size_t)(6 * b + 3) * B + e], cvy = a.cvel[(size_t)(6 * b + 4) * B + e];
  T* out = a.comp + ((size_t)e * 4 + k) * a.ncomp;
  int o = 0;
  for (int j = 0; j < 3; j++) out[o++] = a.prev_action[(size_t)(k * 3 + j) * B + e];
  // corners: home lo / mid / hi, field hi, away hi / mid / lo, field lo; an AWAY player sees them with the goals swapped
  T hl[3], hh[3], al[3], ah[3], fl[2], fh[2];
  for (int i = 0; i < 3; i++) {
    hl[i] = a.lo[(size_t)(0 * 3 + i) * B + e]; hh[i] = a.hi[(size_t)(0 * 3 + i) * B + e];
    al[i] = a.lo[(size_t)(1 * 3 + i) * B + e]; ah[i] = a.hi[(size_t)(1 * 3 + i) * B + e];
  }
  for (int i = 0; i < 2; i++) { fl[i] = a.lo[(size_t)(2 * 3 + i) * B + e]; fh[i] = a.hi[(size_t)(2 * 3 + i) * B + e]; }
  T hm[3], am[3];
  for (int i = 0; i < 3; i++) { hm[i] = (hl[i] + hh[i]) / 2; am[i] = (al[i] + ah[i]) / 2; }
  T corner[8][3] = {{hl[0], hl[1], hl[2]}, {hm[0], hm[1], hm[2]}, {hh[0], hh[1], hh[2]}, {fh[0], fh[1], 0},
                    {ah[0], ah[1], ah[2]}, {am[0], am[1], am[2]}, {al[0], al[1], al[2]}, {fl[0], fl[1], 0}};
  for (int m = 0; m < 8; m++) {
    const int src = away ? (m + 4) & 7 : m;
    const int dim = (m == 1 || m == 5) ? 3 : 2;
    T D[3];
    for (int i = 0; i < 3; i++) D[i] = (corner[src][i] - pos[i]) * (T)((i < 2 || dim == 3) ? 1 : 0);
    for (int j = 0; j < dim; j++) out[o++] = (D[0] * R[j] + D[1] * R[3 + j]) + D[2] * R[6 + j];
  }
  // stats (observables.py:262-375)
  const T dx = ball[0] - pos[0], dy = ball[1] - pos[1], dz = ball[2] - pos[2];
  const T pn = norm2(dx, dy) + (T)1e-7;
  const T vel_to_ball = (dx / pn) * cvx + (dy / pn) * cvy;
  const T dist = norm3(dx, dy, dz);
  const T dmx = ball[0] - pm[0], dmy = ball[1] - pm[1], dmz = ball[2] - pm[2];
  const bool closest = dist <= norm3(dmx, dmy, dmz);
  const T* gm = away ? hm : am;      // the goal the player attacks
  T g[3] = {gm[0] - ball[0], gm[1] - ball[1], gm[2] - ball[2]};
  const T gn = norm3(g[0], g[1], g[2]);
  if (gn > 0) { const T c = gn < (T)1e-30 ? (T)1e-30 : gn; for (int i = 0; i < 3; i++) g[i] = g[i] / c; }
  const T ball_to_goal = (g[0] * blin[0] + g[1] * blin[1]) + g[2] * blin[2];
  const T avg = norm3(pos[0] - pm[0], pos[1] - pm[1], pos[2] - pm[2]);
  out[o++] = vel_to_ball;
  out[o++] = closest ? vel_to_ball : (T)0;
  out[o++] = ball_to_goal;
  out[o++] = avg;
  out[o++] = avg > (T)5 ? (T)1 : (T)0;
  out[o++] = (away ? away_scored : home_scored) ? (T)1 : (T)0;
  out[o++] = (away ? home_scored : away_scored) ? (T)1 : (T)0;
}

extern "C" int soccer_pre(void* stream, const SoccerArgs* a) {
  hipLaunchKernelGGL(soccer_pre_kernel, dim3((a->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}
extern "C" int soccer_post(void* stream, const SoccerArgs* a) {
  hipLaunchKernelGGL(soccer_post_kernel, dim3((4 * a->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}
extern "C" int soccer_args_size() { return (int)sizeof(SoccerArgs); }
