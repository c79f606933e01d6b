// What a wave-level fence (the step kernel's DMC_WSYNC: LDS traffic of ONE wave retires in order) and a workgroup barrier
// between 2 / 4 / 5 waves cost on MI355X, per dependent LDS hand-off -- the price list of "a match on several waves"
// (scripts/multiwave_probe.py).   hipcc --offload-arch=gfx950 -O3 -o barrier_cost_probe barrier_cost_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>      // 0: wave fence, 1: __syncthreads
__global__ void k(float* out, long long* cyc, int iters) {
  __shared__ float buf[512];
  const int t = threadIdx.x, n = blockDim.x;
  float v = (float)t;
  buf[t] = v;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    buf[t] = v;
    if (MODE == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    else __syncthreads();
    v = buf[(t + 1 + (i & 7)) % (MODE == 0 ? 64 : n) + (MODE == 0 ? (t & ~63) : 0)] * 1.0001f + 1.0f;      // a dependent read of another lane's value
    if (MODE == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    else __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * n + t] = v;
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int iters = 20000, blocks = 256;
  float* out; long long* cyc;
  (void)hipMalloc(&out, blocks * 512 * sizeof(float)); (void)hipMalloc(&cyc, blocks * sizeof(long long));
  std::vector<long long> h(blocks);
  for (int waves : {1, 2, 4, 5}) for (int mode : {0, 1}) {
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, iters);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    printf("waves %d  %-14s  %.1f shader cycles per sync point (one trip = LDS write, sync, dependent LDS read, sync: two of them)\n", waves, mode ? "__syncthreads" : "wave fence", s / blocks / iters / 2.0);
  }
  return 0;
}
