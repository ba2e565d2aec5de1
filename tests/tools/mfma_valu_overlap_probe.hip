// Ad-hoc probe (not a test): do MFMA work and plain vector-ALU work of DIFFERENT waves on the same SIMD overlap?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap_probe mfma_valu_overlap_probe.hip && ./mfma_valu_overlap_probe
// One workgroup of 512 threads per CU slot: waves 0-3 (one per SIMD) run v_mfma_f32_16x16x4_f32 on 8 accumulators, waves 4-7
// (the second wave of every SIMD) run v_fma_f32 on 8 accumulators.  Times: both together, MFMA waves alone, VALU waves alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_mix(float* out, int iter_m, int iter_v, float a0, float b0) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  if (wave < 4) {
    if (iter_m > 0) {
      f32x4 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      float a = a0 + threadIdx.x, b = b0;
      for (int it = 0; it < iter_m; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else if (iter_v > 0) {
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (float)i;
    const float a = a0 * 1e-3f + 1.0f, b = b0 * 1e-3f;
    for (int it = 0; it < iter_v; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);       // 32 v_fma_f32 per iteration
    }
    for (int i = 0; i < 8; ++i) s += acc[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float run(float* out, int blocks, int im, int iv) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(512), 0, 0, out, im, iv, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  float* out; hipMalloc(&out, sizeof(float) * 512 * 1024);
  const int blocks = 256;
  // 32 MFMAs = 32 x 32 cycles = 1024 cycles per iteration; 32 v_fma_f32 = 32 x 4 cycles = 128 cycles: 8 VALU iterations per MFMA iteration fill the same time
  const int im = 20000;
  for (int ratio : {2, 4, 6, 8}) {
    const int iv = im * ratio;
    const float tm = run(out, blocks, im, 0), tv = run(out, blocks, 0, iv), tb = run(out, blocks, im, iv);
    printf("VALU iterations per MFMA iteration %d: MFMA alone %.3f ms (%.2f ns per MFMA), VALU alone %.3f ms (%.2f cycles per v_fma_f32 at 2.4 GHz), together %.3f ms = %.2f x (alone + alone)\n",
           ratio, tm, tm * 1e6 / (im * 32.0), tv, tv * 1e-3 * 2.4e9 / (iv * 32.0), tb, tb / (tm + tv));
  }
  return 0;
}
