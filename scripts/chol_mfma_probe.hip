// Probe and device test (VERDICT r05 #1a): the fp32 Cholesky factorisation of the Newton Hessian of a 33 .. 64-dof model, one
// wave per matrix, on the packed lower triangle in LDS -- the two PRODUCTION routines of dm_control_amd/csrc/step_core.h:
//   rows  : chol_factor_rows<N>: lane i keeps row i in N registers, pivots and scaled columns travel by v_readlane
//           (2 instructions per updated entry, N (N - 1) of them)
//   tiles : chol_factor_tiles<N>: blocked right-looking U'U on 16 x 16 tiles in the matrix cores' accumulator layout; a tile
//           fed as BOTH operands of v_mfma_f32_16x16x4_f32, register by register, yields X' Y = the trailing update, with no
//           layout conversion; the columns of a diagonal tile and of its block row are eliminated on the vector ALU
// Prints, for N = 33, 40, 48, 49, 57, 62, 64, cycles per factorisation (s_memtime) with 1 and 5 waves per CU and the error of
// both against an fp64 factor of the same fp32 matrix (tests/test_gpu_chol_tiles.py reads the lines).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chol_probe scripts/chol_mfma_probe.hip && /tmp/chol_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../dm_control_amd/csrc/step_core.h"

#define LDS __attribute__((address_space(3)))

template <int N, int WHICH>
__global__ void __launch_bounds__(320) probe(const float* src, float* out, long long* cycles, int reps) {
  constexpr int NTRI = N * (N + 1) / 2;
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  LDS float* A = (LDS float*)(lds + wave * (NTRI + 64));
  const float* mine = src + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * NTRI;
  long long total = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = lane; i < NTRI + 64; i += 64) A[i] = i < NTRI ? mine[i] : -777.f;      // (the 64 words behind the triangle: a canary)
    DMC_WSYNC();
    const long long t0 = __builtin_readcyclecounter();
    if (WHICH == 0) dmc::chol_factor_rows<float, 64, N>(A, lane); else dmc::chol_factor_tiles<64, N>(A, lane);
    total += __builtin_readcyclecounter() - t0;
  }
  float* o = out + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * (NTRI + 64);
  for (int i = lane; i < NTRI + 64; i += 64) o[i] = A[i];
  if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = total;
}

template <int N> int run() {
  constexpr int NTRI = N * (N + 1) / 2;
  const int nblocks = 256, reps = 20;
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
    hipMalloc(&d_src, h.size() * 4); hipMalloc(&d_out, (size_t)nmat * (NTRI + 64) * 4); hipMalloc(&d_cyc, nmat * 8);
    hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; which++) {
      const size_t lds = (size_t)wpb * (NTRI + 64) * 4;
      if (which == 0) hipLaunchKernelGGL((probe<N, 0>), dim3(nblocks), dim3(64 * wpb), lds, 0, d_src, d_out, d_cyc, reps);
      else hipLaunchKernelGGL((probe<N, 1>), dim3(nblocks), dim3(64 * wpb), lds, 0, d_src, d_out, d_cyc, reps);
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
      std::vector<float> o((size_t)nmat * (NTRI + 64)); std::vector<long long> cyc(nmat);
      hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(cyc.data(), d_cyc, nmat * 8, hipMemcpyDeviceToHost);
      double worst = 0, sum = 0; long long mx = 0; int canary = 0;
      for (int m = 0; m < nmat; m++) {
        for (int i = 0; i < NTRI; i++) { const double r = ref[(size_t)m * NTRI + i], e = fabs(o[(size_t)m * (NTRI + 64) + i] - r) / fmax(1.0, fabs(r)); if (!(e <= worst)) worst = e; }
        for (int i = NTRI; i < NTRI + 64; i++) canary += o[(size_t)m * (NTRI + 64) + i] != -777.f;
        sum += cyc[m]; if (cyc[m] > mx) mx = cyc[m];
      }
      printf("N %d, %d wave(s) per CU, %-5s: %8.0f cycles per factorisation (mean over %d waves; slowest wave %8.0f), max error vs fp64 %.2e, words written past the triangle %d\n",
             N, wpb, which ? "tiles" : "rows", sum / nmat / reps, nmat, (double)mx / reps, worst, canary);
    }
    hipFree(d_src); hipFree(d_out); hipFree(d_cyc);
  }
  return 0;
}

int main() {
  return run<33>() || run<40>() || run<48>() || run<49>() || run<57>() || run<62>() || run<64>();
}
