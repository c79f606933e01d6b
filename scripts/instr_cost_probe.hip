// Cost of a VALU instruction to ONE wave on MI355X (session 7 of round 5): dependent chain vs four independent chains, straight-line
// (128 KB of code) vs a 2 KB loop, the readlane / DPP broadcast patterns of the row routines, an LDS round trip; one wave per CU
// (grid 256) and two per SIMD (grid 2048).  hipcc --offload-arch=gfx950 -O2 scripts/instr_cost_probe.hip -o scripts/session/ifetch
// Measured (profiles/r05_s7_instr_cost_probe.log): dependent v_fma 8.1 - 9.1 cycles, independent 5.1 - 5.9, LDS round trip 74 - 80,
// no difference between the straight-line and the looped form, none between one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R4096(x) R16(R256(x))
// straight-line: 16384 dependent v_fma (VOP3, 8 bytes each = 128 KB of code)
__global__ void straight_dep(float* out, long long* cyc) {
  float a = out[threadIdx.x], b = 1.0001f, c = 0.5f;
  long long t0 = __builtin_readcyclecounter();
  R4(R4096(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));))
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x*64] = a; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// straight-line, 4 independent chains
__global__ void straight_ind(float* out, long long* cyc) {
  float a = out[threadIdx.x], a2 = a + 1, a3 = a + 2, a4 = a + 3, b = 1.0001f, c = 0.5f;
  long long t0 = __builtin_readcyclecounter();
  R4096(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c));)
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x*64] = a + a2 + a3 + a4; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// looped: 256 dependent fma x 64 iterations (2 KB body)
__global__ void loop_dep(float* out, long long* cyc) {
  float a = out[threadIdx.x], b = 1.0001f, c = 0.5f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { R256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x*64] = a; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void loop_ind(float* out, long long* cyc) {
  float a = out[threadIdx.x], a2 = a + 1, a3 = a + 2, a4 = a + 3, b = 1.0001f, c = 0.5f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { R16(R4(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c));)) }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x*64] = a + a2 + a3 + a4; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// mixed pattern like the step kernel: v_mul, s_nop 1, v_mov_dpp, v_fma (dependent), straight-line 4096 groups
__global__ void straight_mix(float* out, long long* cyc) {
  float a = out[threadIdx.x], b = 1.0001f, t;
  long long t0 = __builtin_readcyclecounter();
  R4096(asm volatile("v_mul_f32 %1, %0, %2\n s_nop 1\n v_mov_b32_dpp %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %0, %1, %2, %0" : "+v"(a), "=&v"(t) : "v"(b));)
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x*64] = a; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// readlane pattern: v_readlane x2, cndmask, fma
__global__ void straight_rl(float* out, long long* cyc) {
  float a = out[threadIdx.x], b = 1.0001f, t, t2;
  int s1, s2;
  long long t0 = __builtin_readcyclecounter();
  R4096(asm volatile("v_readlane_b32 %2, %0, 3\n v_readlane_b32 %3, %0, 35\n v_mov_b32 %1, %2\n v_mov_b32 %5, %3\n v_cndmask_b32 %1, %1, %5, vcc\n v_fma_f32 %0, %1, %4, %0" : "+v"(a), "=&v"(t), "=&s"(s1), "=&s"(s2) : "v"(b), "v"(t2) : "vcc");)
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x*64] = a; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// LDS round trip chain: write, read, dependent
__global__ void lds_chain(float* out, long long* cyc) {
  __shared__ float sh[64];
  float a = out[threadIdx.x];
  sh[threadIdx.x] = a;
  long long t0 = __builtin_readcyclecounter();
  int idx = threadIdx.x;
  for (int i = 0; i < 1024; i++) { float v = sh[idx & 63]; idx = (int)(v) + idx + 1; }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x*64] = idx; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename K> void run(const char* name, K k, int grid, double ninstr, float* d, long long* c) {
  static long long h[4096];
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, c); hipDeviceSynchronize();
    hipMemcpy(h, c, grid*8, hipMemcpyDeviceToHost);
    long long mn = h[0], mx = h[0]; double s = 0; for (int i = 0; i < grid; i++) { if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; s += h[i]; }
    printf("%-14s grid %4d rep %d: cycles/instr min %.2f mean %.2f max %.2f\n", name, grid, rep, mn/ninstr, s/grid/ninstr, mx/ninstr);
  }
}
int main() {
  float* d; long long* c; hipMalloc(&d, 4096*64*4); hipMemset(d, 0, 4096*64*4); hipMalloc(&c, 4096*8);
  for (int grid : {256, 2048}) {
    run("straight_dep", straight_dep, grid, 16384, d, c);
    run("straight_ind", straight_ind, grid, 16384, d, c);
    run("loop_dep", loop_dep, grid, 16384, d, c);
    run("loop_ind", loop_ind, grid, 16384, d, c);
    run("straight_mix", straight_mix, grid, 4096*4, d, c);
    run("straight_rl", straight_rl, grid, 4096*6, d, c);
    run("lds_chain", lds_chain, grid, 1024, d, c);
  }
  return 0;
}
