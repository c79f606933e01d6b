// Probe (VERDICT r05 #1a): the 62 x 62 fp32 Cholesky factorisation of the Newton Hessian, one wave per matrix, on the packed
// lower triangle in LDS that step_core.h's chol_factor_rows<62> works on --
//   rows  : lane i keeps row i in 62 registers, pivots and scaled columns travel by v_readlane (the production routine,
//           restated: 2 instructions per updated entry, N (N - 1) of them)
//   mfma  : blocked right-looking U'U on 16 x 16 tiles in the matrix cores' accumulator layout (lane 16 g + c, register r
//           holds element (4 g + r, c)); a tile fed as BOTH operands of v_mfma_f32_16x16x4_f32, register by register,
//           yields X' Y, which is exactly the trailing update  A_ij -= U_ki' U_kj  of the upper-triangular form -- no
//           layout conversion.  The 16 columns of a diagonal tile and of its block row are eliminated with DPP
//           row_newbcast (the column, by symmetry of the tile) and one cross-row-group broadcast (the row) per column.
// Prints cycles per factorisation (s_memtime) with 1 and 5 waves per CU and the error of both against an fp64 factor.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chol_probe scripts/chol_mfma_probe.hip && /tmp/chol_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define MINVAL 1e-15f
constexpr int N = 62, NB = (N + 15) / 16, NTRI = N * (N + 1) / 2;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define LDS __attribute__((address_space(3)))

__device__ __forceinline__ constexpr int tri_c0(int j) { return j * N - ((j * (j - 1)) >> 1); }
__device__ __forceinline__ float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
template <int CTRL> __device__ __forceinline__ float dpp_all(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// ---- production form ------------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) void chol_rows(LDS float* A, int lane) {
  float a[N];
  const bool own = lane < N;
#pragma unroll
  for (int j = 0; j < N; j++) a[j] = (own && j <= lane) ? A[tri_c0(j) + lane - j] : 0.f;
#pragma unroll
  for (int k = 0; k < N; k++) {
    float akk = readlane_f(a[k], k);
    if (akk < MINVAL) akk = MINVAL;
    const float inv = rsq(akk);
    const float lik = a[k] * inv;
#pragma unroll
    for (int j = k + 1; j < N; j++) { const float ljk = readlane_f(lik, j); a[j] = a[j] - lik * ljk; }
    a[k] = lane == k ? inv : lik;
  }
#pragma unroll
  for (int j = 0; j < N; j++) if (own && j <= lane) A[tri_c0(j) + lane - j] = a[j];
}

// ---- matrix-core form -----------------------------------------------------------------------------------------------
#ifndef BCAST_BPERMUTE
#define BCAST_BPERMUTE 1
#endif
// the value the lane of the same column in row group GC holds, for every row group
template <int GC, bool DIAG> __device__ __forceinline__ float bcast_rowgroup(float x, int col4) {
#if BCAST_BPERMUTE == 1
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(col4 + 64 * GC, __builtin_bit_cast(int, x)));
#elif BCAST_BPERMUTE == 2      // the diagonal tile (the critical chain) by swaps, the block row's other tiles through the crossbar
  if (!DIAG) return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(col4 + 64 * GC, __builtin_bit_cast(int, x)));
  const unsigned u = __float_as_uint(x);
  const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned y = GC < 2 ? h[0] : h[1];
  const auto q = __builtin_amdgcn_permlane16_swap(y, y, false, false);
  return __uint_as_float((GC & 1) ? q[1] : q[0]);
#else
  const unsigned u = __float_as_uint(x);
  const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // h[0]: rows (0 1 0 1), h[1]: rows (2 3 2 3)
  const unsigned y = GC < 2 ? h[0] : h[1];
  const auto q = __builtin_amdgcn_permlane16_swap(y, y, false, false);      // q[0]: even rows everywhere, q[1]: odd rows
  return __uint_as_float((GC & 1) ? q[1] : q[0]);
#endif
}
template <int C0> __device__ __forceinline__ float col_bcast(float v) { return dpp_all<0x150 + C0>(v); }

struct TileLane { int g, col4, lane; float fgt[3]; };      // fgt[q] = 1 where the lane's row group g > q, else 0

template <int K, int C0>
__device__ __forceinline__ void eliminate_column(f4 (&t)[NB][NB], const TileLane& tl) {
  constexpr int GC = C0 >> 2, RC = C0 & 3;
  f4& T = t[K][K];
  const float inv = rsq(__builtin_amdgcn_fmed3f(readlane_f(T[RC], 16 * GC + C0), MINVAL, __builtin_inff()));
  // the column below the pivot, scaled -- by symmetry of the diagonal tile, column C0 of the lane's own row group; rows at
  // or above the pivot row get a zero multiplier (folded into the scale: 4 g + r > C0  <=>  r > RC ? g >= GC : g > GC)
  const float inv_ge = GC == 0 ? inv : inv * tl.fgt[GC > 0 ? GC - 1 : 0];
  const float inv_gt = GC == 3 ? 0.f : inv * tl.fgt[GC < 3 ? GC : 0];
  float ui[4];
#pragma unroll
  for (int r = 0; r < 4; r++) ui[r] = (GC == 3 && r <= RC) ? 0.f : col_bcast<C0>(T[r]) * (r > RC ? inv_ge : inv_gt);
  const float scale = tl.g == GC ? inv : 1.f;      // the pivot row itself is scaled in place
#pragma unroll
  for (int j = K; j < NB; j++) {
    f4& P = t[K][j];
    P[RC] = P[RC] * scale;
    const float X = j == K ? bcast_rowgroup<GC, true>(P[RC], tl.col4) : bcast_rowgroup<GC, false>(P[RC], tl.col4);      // the scaled pivot row, in every row group
#pragma unroll
    for (int r = 0; r < 4; r++) if (!(GC == 3 && r <= RC)) P[r] = P[r] - ui[r] * X;      // (as v_pk_fma_f32 pairs: measured slower -- the multipliers then travel through v_mov_dpp + v_pk_mul)
  }
  T[RC] = (tl.lane == 16 * GC + C0) ? inv : T[RC];      // the packed form keeps 1 / L_kk on the diagonal
}
template <int K, int C0> struct Columns {
  static __device__ __forceinline__ void run(f4 (&t)[NB][NB], const TileLane& tl) {
    eliminate_column<K, C0>(t, tl);
    if constexpr (C0 + 1 < 16) Columns<K, C0 + 1>::run(t, tl);
  }
};
template <int K> __device__ __forceinline__ void block_column(f4 (&t)[NB][NB], const TileLane& tl) {
  Columns<K, 0>::run(t, tl);
#pragma unroll
  for (int i = K + 1; i < NB; i++) {
    const f4 nx = -t[K][i];
#pragma unroll
    for (int j = i; j < NB; j++) {
#pragma unroll
      for (int r = 0; r < 4; r++) t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(nx[r], t[K][j][r], t[i][j], 0, 0, 0);
    }
  }
  if constexpr (K + 1 < NB) block_column<K + 1>(t, tl);
}

// Packed index of element (R, C), R <= C, of tile (bi, bj), register r, for the lane (g, c):
//   tri_c0(R) + C - R  with  R = R0 + G (R0 = 16 bi + r, G = 4 g)  =  [tri_c0(R0) + 16 bj - R0] + [tri_c0(G) - G + c] - R0 G
__device__ __attribute__((noinline)) void chol_mfma(LDS float* A, int lane) {
  const int g = lane >> 4, c = lane & 15, G = 4 * g;
  TileLane tl; tl.g = g; tl.col4 = 4 * c; tl.lane = lane;
#pragma unroll
  for (int q = 0; q < 3; q++) tl.fgt[q] = g > q ? 1.f : 0.f;
  const int up = tri_c0(G) - G + c;                  // lane part of the upper-triangle index
  const int tc = ((c * (2 * N + 1 - c)) >> 1) - c;   // tri_c0(c) - c: lane part of the mirrored (lower-triangle) index
  f4 t[NB][NB];
#pragma unroll
  for (int bi = 0; bi < NB; bi++)
#pragma unroll
    for (int bj = bi; bj < NB; bj++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int R0 = 16 * bi + r;
        int at = tri_c0(R0) + 16 * bj - R0 + up - R0 * G;
        if (bi == bj) {      // the diagonal tiles enter whole (symmetric): below the diagonal the mirrored entry
          const int low = tri_c0(16 * bi) + tc - 16 * bi * c + R0 - 16 * bi + G;
          at = (G + r <= c) ? at : low;
        }
        float v = A[at];
        // rows / columns past N: the identity (its factor is the identity; nothing of it is stored)
        if (16 * bj + 15 >= N) { const bool in = (16 * bj + c < N) && (bi < bj || 16 * bi + G + r < N); v = in ? v : ((bi == bj && G + r == c) ? 1.f : 0.f); }
        t[bi][bj][r] = v;
      }
  block_column<0>(t, tl);
  // stores: an entry that does not exist (below the diagonal of a diagonal tile, past column N) aims at the slot of
  // the lane's entry (row 4 g, column 16 + c) -- which exists for every lane and is stored LAST, over whatever landed there
  const int safe = tri_c0(0) + 16 - 0 + up - 0 * G;
#pragma unroll
  for (int bi = NB - 1; bi >= 0; bi--)
#pragma unroll
    for (int bj = NB - 1; bj >= bi; bj--)
#pragma unroll
      for (int r = 3; r >= 0; r--) {
        const int R0 = 16 * bi + r;
        int at = tri_c0(R0) + 16 * bj - R0 + up - R0 * G;
        bool ok = true;
        if (bi == bj) ok = G + r <= c;
        if (16 * bj + 15 >= N) ok = ok && (16 * bj + c < N);
        if (bi == bj || 16 * bj + 15 >= N) at = ok ? at : safe;
        if (!(bi == 0 && bj == 1 && r == 0)) A[at] = t[bi][bj][r];
      }
  A[safe] = t[0][1][0];
}

template <int WHICH>
__global__ void __launch_bounds__(320) probe(const float* src, float* out, long long* cycles, int reps) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  LDS float* A = (LDS float*)(lds + wave * (NTRI + 64));
  const float* mine = src + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * NTRI;
  long long total = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = lane; i < NTRI; i += 64) A[i] = mine[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const long long t0 = __builtin_readcyclecounter();
    if (WHICH == 0) chol_rows(A, lane); else chol_mfma(A, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    total += __builtin_readcyclecounter() - t0;
  }
  float* o = out + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * NTRI;
  for (int i = lane; i < NTRI; i += 64) o[i] = A[i];
  if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = total;
}

int main() {
  const int nblocks = 256, reps = 50;
  for (int wpb : {1, 5}) {
    const int nmat = nblocks * wpb;
    std::vector<float> h((size_t)nmat * NTRI);
    std::vector<double> ref((size_t)nmat * NTRI);
    srand(1);
    for (int m = 0; m < nmat; m++) {      // H = M + J' D J: a diagonally heavy SPD matrix with a dense low-rank part
      std::vector<double> G(N * N, 0.0), J(20 * N);
      for (auto& x : J) x = (rand() / (double)RAND_MAX - 0.5);
      for (int i = 0; i < N; i++) for (int j = 0; j <= i; j++) {
        double s = i == j ? 1.0 + 3.0 * rand() / RAND_MAX : ((abs(i - j) < 6) ? 0.2 * (rand() / (double)RAND_MAX - 0.5) : 0.0);
        for (int r = 0; r < 20; r++) s += 5.0 * J[r * N + i] * J[r * N + j];
        G[i * N + j] = s;
      }
      for (int j = 0; j < N; j++) for (int i = j; i < N; i++) h[(size_t)m * NTRI + j * N - j * (j - 1) / 2 + i - j] = (float)G[i * N + j];
      // fp64 factor of the fp32-rounded matrix, in the packed form (scaled columns, 1 / L_kk on the diagonal)
      std::vector<double> Lm(N * N, 0.0);
      for (int j = 0; j < N; j++) for (int i = j; i < N; i++) Lm[i * N + j] = (double)h[(size_t)m * NTRI + j * N - j * (j - 1) / 2 + i - j];
      for (int k = 0; k < N; k++) {
        const double inv = 1.0 / sqrt(Lm[k * N + k]);
        for (int i = k + 1; i < N; i++) Lm[i * N + k] *= inv;
        for (int j = k + 1; j < N; j++) for (int i = j; i < N; i++) Lm[i * N + j] -= Lm[i * N + k] * Lm[j * N + k];
        Lm[k * N + k] = inv;
      }
      for (int j = 0; j < N; j++) for (int i = j; i < N; i++) ref[(size_t)m * NTRI + j * N - j * (j - 1) / 2 + i - j] = Lm[i * N + j];
    }
    float *d_src, *d_out; long long* d_cyc;
    hipMalloc(&d_src, h.size() * 4); hipMalloc(&d_out, h.size() * 4); hipMalloc(&d_cyc, nmat * 8);
    hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; which++) {
      const size_t lds = (size_t)wpb * (NTRI + 64) * 4;
      if (which == 0) hipLaunchKernelGGL(probe<0>, dim3(nblocks), dim3(64 * wpb), lds, 0, d_src, d_out, d_cyc, reps);
      else hipLaunchKernelGGL(probe<1>, dim3(nblocks), dim3(64 * wpb), lds, 0, d_src, d_out, d_cyc, reps);
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
      std::vector<float> o(h.size()); std::vector<long long> cyc(nmat);
      hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(cyc.data(), d_cyc, nmat * 8, hipMemcpyDeviceToHost);
      double worst = 0, sum = 0; long long mx = 0;
      for (size_t i = 0; i < o.size(); i++) { const double e = fabs(o[i] - ref[i]) / fmax(1.0, fabs(ref[i])); if (!(e <= worst)) worst = e; }
      for (int m = 0; m < nmat; m++) { sum += cyc[m]; if (cyc[m] > mx) mx = cyc[m]; }
      printf("%d wave(s) per CU, %-5s: %8.0f cycles per factorisation (mean over %d waves; slowest wave %8.0f), max error vs fp64 %.2e\n",
             wpb, which ? "mfma" : "rows", sum / nmat / reps, nmat, (double)mx / reps, worst);
    }
    hipFree(d_src); hipFree(d_out); hipFree(d_cyc);
  }
  return 0;
}
