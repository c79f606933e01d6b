// Device unit test of the CROSS-LANE layer of dm_control_amd/csrc/step_core.h, in isolation from the step kernel: the
// reductions / broadcasts / scans (DPP quad_perm / row_mirror / row_shr / row_newbcast / row_bcast, v_permlane16/32_swap,
// v_readlane) for 16, 32 and 64 lanes per environment, and the row-per-lane linear algebra built on them
// (chol_factor_rows / chol_solve_rows, the per-tree chol_factor_trees / chol_solve_trees) against host references and
// against the fenced LDS forms (chol_factor_lds / chol_solve_lds) they replaced.  None of this is reachable on the CPU
// tier: tests/emu runs the kernel core with one lane per environment, where every one of these is the identity.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/lane_probe scripts/lane_primitives_probe.hip && /tmp/lane_probe
// One line per check: "<name> lpe <L> n <N> <type>: max_err <e> mismatches <k>"; tests/test_gpu_lane_primitives.py reads them.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../dm_control_amd/csrc/step_core.h"

#define LDS __attribute__((address_space(3)))
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static int failures = 0;

// ---- reductions, broadcasts, scans ----------------------------------------------------------------------------------
// out: per lane [sum_f, sum_d (as two floats hi/lo is overkill: written as double array), ...]
template <int LPE>
__global__ void prim_kernel(const float* vf, const double* vd, const int* vi, float* of, double* od, int* oi) {
  const int l = threadIdx.x;      // one wave
  const int lane = l % LPE;
  of[l] = dmc::group_sum<LPE>(vf[l]);
  od[l] = dmc::group_sum<LPE>(vd[l]);
  oi[l] = 0;
  oi[64 + l] = dmc::group_max<LPE>(vi[l]);
  int total;
  oi[128 + l] = dmc::group_scan<LPE>(vi[l] & 7, lane, &total);
  oi[192 + l] = total;
  // wave-uniform broadcasts of lane k of each group, k = 0 .. LPE - 1 (unrolled: bcast_rows wants a constant)
#pragma unroll
  for (int k = 0; k < LPE; k++) {
    of[64 * (1 + k) + l] = dmc::wave_bcast<LPE>(vf[l], k);
    od[64 * (1 + k) + l] = dmc::wave_bcast<LPE>(vd[l], k);
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {      // the row_newbcast form: valid for the lanes of each group's FIRST 16-lane row
    of[64 * (65 + k) + l] = dmc::bcast_rows<LPE, 16>(vf[l], k);
    od[64 * (65 + k) + l] = dmc::bcast_rows<LPE, 16>(vd[l], k);
  }
}

template <int LPE> void run_primitives() {
  std::vector<float> vf(64); std::vector<double> vd(64); std::vector<int> vi(64);
  srand(7 + LPE);
  for (int l = 0; l < 64; l++) { vf[l] = (float)(rand() % 2001 - 1000) / 8.f; vd[l] = (rand() % 2000001 - 1000000) / 1024.0; vi[l] = rand() % 100000 - 50000; }
  float *dvf, *of; double *dvd, *od; int *dvi, *oi;
  const size_t NF = 64 * 81;
  CHECK(hipMalloc(&dvf, 64 * 4)); CHECK(hipMalloc(&dvd, 64 * 8)); CHECK(hipMalloc(&dvi, 64 * 4));
  CHECK(hipMalloc(&of, NF * 4)); CHECK(hipMalloc(&od, NF * 8)); CHECK(hipMalloc(&oi, 256 * 4));
  CHECK(hipMemcpy(dvf, vf.data(), 64 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dvd, vd.data(), 64 * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dvi, vi.data(), 64 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(of, 0, NF * 4)); CHECK(hipMemset(od, 0, NF * 8));
  hipLaunchKernelGGL(prim_kernel<LPE>, dim3(1), dim3(64), 0, 0, dvf, dvd, dvi, of, od, oi);
  CHECK(hipDeviceSynchronize());
  std::vector<float> hf(NF); std::vector<double> hd(NF); std::vector<int> hi(256);
  CHECK(hipMemcpy(hf.data(), of, NF * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hd.data(), od, NF * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hi.data(), oi, 256 * 4, hipMemcpyDeviceToHost));
  int bad_sum = 0, bad_max = 0, bad_scan = 0, bad_bcast = 0, bad_rowb = 0;
  for (int l = 0; l < 64; l++) {
    const int g0 = (l / LPE) * LPE;
    double sf = 0, sd = 0; int mx = -1 << 30, pre = 0, tot = 0;
    for (int j = g0; j < g0 + LPE; j++) { sf += vf[j]; sd += vd[j]; mx = vi[j] > mx ? vi[j] : mx; if (j < l) pre += vi[j] & 7; tot += vi[j] & 7; }
    // (the inputs are multiples of 1/8 and 1/1024 below 2^24 / 2^53: every partial sum is exact, whatever the pairing tree)
    if ((double)hf[l] != sf || hd[l] != sd) bad_sum++;
    if (hi[64 + l] != mx) bad_max++;
    if (hi[128 + l] != pre || hi[192 + l] != tot) bad_scan++;
    for (int k = 0; k < LPE; k++) if (hf[64 * (1 + k) + l] != vf[g0 + k] || hd[64 * (1 + k) + l] != vd[g0 + k]) bad_bcast++;
    if (LPE >= 16 && l - g0 < 16) for (int k = 0; k < 16; k++) if (hf[64 * (65 + k) + l] != vf[g0 + k] || hd[64 * (65 + k) + l] != vd[g0 + k]) bad_rowb++;
  }
  printf("group_sum lpe %d n 0 f32+f64: max_err 0 mismatches %d\n", LPE, bad_sum);
  printf("group_max lpe %d n 0 i32: max_err 0 mismatches %d\n", LPE, bad_max);
  printf("group_scan lpe %d n 0 i32: max_err 0 mismatches %d\n", LPE, bad_scan);
  printf("wave_bcast lpe %d n 0 f32+f64: max_err 0 mismatches %d\n", LPE, bad_bcast);
  if (LPE >= 16) printf("bcast_rows16 lpe %d n 16 f32+f64: max_err 0 mismatches %d\n", LPE, bad_rowb);
  failures += bad_sum + bad_max + bad_scan + bad_bcast + bad_rowb;
  hipFree(dvf); hipFree(dvd); hipFree(dvi); hipFree(of); hipFree(od); hipFree(oi);
}

// ---- row-per-lane Cholesky / substitution, 64 / LPE environments per wave --------------------------------------------
// src: per group the packed lower triangle (by columns) then the right-hand side; out: factor (rows form), x (rows form),
// factor (LDS form), x (LDS form)
template <typename T, int LPE, int N>
__global__ void chol_kernel(const T* src, T* out) {
  constexpr int NTRI = N * (N + 1) / 2, SLAB = NTRI + 2 * N + 8;
  extern __shared__ unsigned char raw[];
  const int l = threadIdx.x, g = l / LPE, lane = l % LPE;
  LDS T* A = (LDS T*)(T*)raw + g * SLAB;
  LDS T* b = A + NTRI;
  LDS T* x = b + N;
  const T* mine = src + (size_t)g * (NTRI + N);
  T* o = out + (size_t)g * 2 * (NTRI + N);
  for (int form = 0; form < 2; form++) {
    for (int i = lane; i < NTRI + N; i += LPE) A[i] = mine[i];
    DMC_WSYNC();
    if (form == 0) { dmc::chol_factor_rows<T, LPE, N>(A, lane); dmc::chol_solve_rows<T, LPE, N>(x, A, b, lane); }
    else { dmc::chol_factor_lds<T, LPE>(A, N, lane); dmc::chol_solve_lds<T, LPE>(x, A, b, N, lane); }
    DMC_WSYNC();
    for (int i = lane; i < NTRI; i += LPE) o[form * (NTRI + N) + i] = A[i];
    for (int i = lane; i < N; i += LPE) o[form * (NTRI + N) + NTRI + i] = x[i];
    DMC_WSYNC();
  }
}

// block-diagonal N x N (NT trees of TM dofs): the per-tree routines against the whole-matrix row routines
template <typename T, int N, int TM>
__global__ void trees_kernel(const T* src, T* out) {
  constexpr int NTRI = N * (N + 1) / 2;
  extern __shared__ unsigned char raw[];
  const int lane = threadIdx.x;
  LDS T* A = (LDS T*)(T*)raw;
  LDS T* b = A + NTRI;
  LDS T* x = b + N;
  const int t0 = lane < N ? (lane / TM) * TM : 0, t1 = lane < N ? t0 + TM : 0;
  for (int form = 0; form < 2; form++) {
    for (int i = lane; i < NTRI + N; i += 64) A[i] = src[i];
    DMC_WSYNC();
    if (form == 0) { dmc::chol_factor_rows<T, 64, N>(A, lane); dmc::chol_solve_rows<T, 64, N>(x, A, b, lane); }
    else { dmc::chol_factor_trees<T, 64, N, TM>(A, lane, t0, t1); dmc::chol_solve_trees<T, 64, N, TM>(x, A, b, lane, t0, t1); }
    DMC_WSYNC();
    for (int i = lane; i < NTRI; i += 64) out[form * (NTRI + N) + i] = A[i];
    for (int i = lane; i < N; i += 64) out[form * (NTRI + N) + NTRI + i] = x[i];
    DMC_WSYNC();
  }
}

static void spd(int N, int block, std::vector<double>& G, std::vector<double>& rhs) {
  // H = M + J' D J restricted to blocks of `block` dofs (block = N: dense)
  G.assign(N * N, 0.0); rhs.resize(N);
  std::vector<double> J(12 * N);
  for (auto& v : J) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < N; i++) for (int j = 0; j <= i; j++) {
    if (i / block != j / block) continue;
    double s = i == j ? 1.0 + 3.0 * rand() / RAND_MAX : 0.2 * (rand() / (double)RAND_MAX - 0.5);
    for (int r = 0; r < 12; r++) s += 4.0 * J[r * N + i] * J[r * N + j];
    G[i * N + j] = G[j * N + i] = s;
  }
  for (int i = 0; i < N; i++) rhs[i] = 10.0 * (rand() / (double)RAND_MAX - 0.5);
}

template <typename T> static void pack(int N, const std::vector<double>& G, const std::vector<double>& rhs, T* dst) {
  int p = 0;
  for (int j = 0; j < N; j++) for (int i = j; i < N; i++) dst[p++] = (T)G[i * N + j];
  for (int i = 0; i < N; i++) dst[p++] = (T)rhs[i];
}

// fp64 solve of the matrix AS ROUNDED to T
template <typename T> static void ref_solve(int N, const T* packed, std::vector<double>& x) {
  std::vector<double> L(N * N, 0.0), y(N);
  int p = 0;
  for (int j = 0; j < N; j++) for (int i = j; i < N; i++) L[i * N + j] = (double)packed[p++];
  for (int k = 0; k < N; k++) {
    L[k * N + k] = sqrt(L[k * N + k]);
    for (int i = k + 1; i < N; i++) L[i * N + k] /= L[k * N + k];
    for (int j = k + 1; j < N; j++) for (int i = j; i < N; i++) L[i * N + j] -= L[i * N + k] * L[j * N + k];
  }
  for (int i = 0; i < N; i++) { double s = (double)packed[p + i]; for (int k = 0; k < i; k++) s -= L[i * N + k] * y[k]; y[i] = s / L[i * N + i]; }
  x.resize(N);
  for (int i = N - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < N; k++) s -= L[k * N + i] * x[k]; x[i] = s / L[i * N + i]; }
}

template <typename T, int LPE, int N> void run_chol(const char* tname) {
  constexpr int NTRI = N * (N + 1) / 2, SLAB = NTRI + 2 * N + 8, G = 64 / LPE;
  std::vector<T> h((size_t)G * (NTRI + N)), o((size_t)G * 2 * (NTRI + N));
  std::vector<std::vector<double>> want(G);
  srand(100 * LPE + N);
  for (int g = 0; g < G; g++) {
    std::vector<double> M, r; spd(N, N, M, r);
    pack<T>(N, M, r, h.data() + (size_t)g * (NTRI + N));
    ref_solve<T>(N, h.data() + (size_t)g * (NTRI + N), want[g]);
  }
  T *d, *dout;
  CHECK(hipMalloc(&d, h.size() * sizeof(T))); CHECK(hipMalloc(&dout, o.size() * sizeof(T)));
  CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  hipLaunchKernelGGL((chol_kernel<T, LPE, N>), dim3(1), dim3(64), G * SLAB * sizeof(T), 0, d, dout);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(o.data(), dout, o.size() * sizeof(T), hipMemcpyDeviceToHost));
  double err = 0; int differ = 0;
  for (int g = 0; g < G; g++) {
    const T* rows = o.data() + (size_t)g * 2 * (NTRI + N), *lds = rows + NTRI + N;
    double scale = 0;
    for (int i = 0; i < N; i++) scale = fmax(scale, fabs(want[g][i]));
    for (int i = 0; i < N; i++) { err = fmax(err, fabs((double)rows[NTRI + i] - want[g][i]) / scale); err = fmax(err, fabs((double)lds[NTRI + i] - want[g][i]) / scale); }
    // "same arithmetic per entry, in the same order": the register form against the fenced LDS form, bit for bit
    if (memcmp(rows, lds, (NTRI + N) * sizeof(T))) for (int i = 0; i < NTRI + N; i++) if (memcmp(&rows[i], &lds[i], sizeof(T))) differ++;
  }
  printf("chol_rows_vs_fp64_and_lds lpe %d n %d %s: max_err %.3e mismatches %d\n", LPE, N, tname, err, differ);
  const double tol = sizeof(T) == 4 ? 2e-4 : 1e-11;
  if (!(err < tol)) failures++;
  hipFree(d); hipFree(dout);
}

template <typename T, int N, int TM> void run_trees(const char* tname) {
  constexpr int NTRI = N * (N + 1) / 2;
  std::vector<T> h(NTRI + N), o(2 * (NTRI + N));
  srand(1000 + N);
  std::vector<double> M, r, want; spd(N, TM, M, r);
  pack<T>(N, M, r, h.data());
  ref_solve<T>(N, h.data(), want);
  T *d, *dout;
  CHECK(hipMalloc(&d, h.size() * sizeof(T))); CHECK(hipMalloc(&dout, o.size() * sizeof(T)));
  CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  hipLaunchKernelGGL((trees_kernel<T, N, TM>), dim3(1), dim3(64), (NTRI + 2 * N + 8) * sizeof(T), 0, d, dout);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(o.data(), dout, o.size() * sizeof(T), hipMemcpyDeviceToHost));
  double err = 0, scale = 0; int differ = 0;
  for (int i = 0; i < N; i++) scale = fmax(scale, fabs(want[i]));
  const T* rows = o.data(), *trees = o.data() + NTRI + N;
  for (int i = 0; i < N; i++) err = fmax(err, fabs((double)trees[NTRI + i] - want[i]) / scale);
  // in-tree entries and the solution: bit-identical to the whole-matrix routines (what is skipped is a - 0 * x)
  int p = 0;
  for (int j = 0; j < N; j++) for (int i = j; i < N; i++, p++) if (i / TM == j / TM && memcmp(&rows[p], &trees[p], sizeof(T))) differ++;
  for (int i = 0; i < N; i++) if (memcmp(&rows[NTRI + i], &trees[NTRI + i], sizeof(T))) differ++;
  printf("chol_trees_vs_rows lpe 64 n %d %s: max_err %.3e mismatches %d\n", N, tname, err, differ);
  const double tol = sizeof(T) == 4 ? 2e-4 : 1e-11;
  if (!(err < tol) || differ) failures++;
  hipFree(d); hipFree(dout);
}

int main() {
  run_primitives<16>(); run_primitives<32>(); run_primitives<64>();
  run_chol<float, 16, 3>("f32"); run_chol<float, 16, 9>("f32"); run_chol<float, 16, 16>("f32");
  run_chol<float, 32, 9>("f32"); run_chol<float, 32, 17>("f32"); run_chol<float, 32, 27>("f32");
  run_chol<float, 64, 9>("f32"); run_chol<float, 64, 27>("f32"); run_chol<float, 64, 30>("f32"); run_chol<float, 64, 62>("f32");
  run_chol<double, 16, 9>("f64"); run_chol<double, 32, 9>("f64"); run_chol<double, 32, 27>("f64"); run_chol<double, 64, 27>("f64"); run_chol<double, 64, 62>("f64");
  run_trees<float, 30, 6>("f32"); run_trees<double, 30, 6>("f64"); run_trees<float, 24, 8>("f32");
  printf("failures %d\n", failures);
  return failures ? 1 : 0;
}
