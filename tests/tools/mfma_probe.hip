// Ad-hoc micro-benchmark (not part of the product): fp32 MFMA issue rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + u, b + n, acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + u, b + n, acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern, int blocks_per_cu, double flop_per_mfma, int nacc) {
  float* d; hipMalloc(&d, 256 * 256 * 8 * 4 * sizeof(float));
  const int iters = 4000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 256>>>(d, 10, 1.f, 2.f);
  hipEventRecord(e0);
  kern<<<grid, 256>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)grid * 4 * iters * 8 * nacc;
  printf("%-28s blocks/CU=%d  %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.2GHz)\n", name, blocks_per_cu, ms, n * flop_per_mfma / ms / 1e9,
         ms * 1e-3 * 2.2e9 / (n / 1024.0));
  hipFree(d);
}
int main() {
  for (int b : {1, 3}) {
    run("32x32x2 f32, 1 acc", k32<1>, b, 4096, 1);
    run("32x32x2 f32, 2 acc", k32<2>, b, 4096, 2);
    run("32x32x2 f32, 4 acc", k32<4>, b, 4096, 4);
    run("16x16x4 f32, 2 acc", k16<2>, b, 2048, 2);
    run("16x16x4 f32, 8 acc", k16<8>, b, 2048, 8);
  }
  return 0;
}
