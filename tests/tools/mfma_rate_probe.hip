// Ad-hoc probe (not a test): sustained issue rate of v_mfma_f32_16x16x4_f32 under the operand traffic of the filter bank's K loop.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe mfma_rate_probe.hip && ./mfma_rate_probe
// Every wave runs ITER blocks of 32 MFMAs on 8 independent accumulators, operands in 16 A + 8 B registers (random data).
// Template switches: DS = four ds_read_b128 per block (A of the next block); GL = B of the next block from global memory as
// 1: eight global_load_dword, 2: two global_load_dwordx4; IL = loads interleaved between the MFMAs instead of ahead of them.
// Reported: ns per MFMA per SIMD from HIP events (13.3 ns = 32 cycles at 2.4 GHz = peak), clock inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int DS, int GL, int IL>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* clk, int iter, const float* __restrict__ rnd) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a[4], an[4];
  f32x4 b[2], bn[2];
  for (int m = 0; m < 4; ++m) a[m] = an[m] = *(const f32x4*)(rnd + threadIdx.x * 16 + 4 * m);
  for (int i = 0; i < 2; ++i) b[i] = bn[i] = *(const f32x4*)(rnd + 4096 + threadIdx.x * 8 + 4 * i);
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = rnd[i];
  __syncthreads();
  const float* g = rnd + 8192 + 4 * (threadIdx.x & 63);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  auto ld_a = [&](int m, int it) { if (DS) an[m] = *(const f32x4*)(lds + ((threadIdx.x * 4 + 1024 * m + 64 * (it & 7)) & 4092)); };
  auto ld_b = [&](int i, int it) {   // i = 0..7
    if (GL == 1) bn[i >> 2][i & 3] = g[(size_t)((it & 31) * 8 + i) * 260];
    if (GL == 2 && (i & 3) == 0) bn[i >> 2] = *(const f32x4*)(g + (size_t)((it & 31) * 2 + (i >> 2)) * 260);
  };
  for (int it = 0; it < iter; ++it) {
    if (!IL) {
#pragma unroll
      for (int i = 0; i < 8; ++i) ld_b(i, it);
#pragma unroll
      for (int m = 0; m < 4; ++m) ld_a(m, it);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3][u], b[i >> 2][u], acc[i], 0, 0, 0);
        if (IL) {
          const int q = 8 * u + i;
          if (q % 4 == 1 && q / 4 < 8) ld_b(q / 4, it);
          if (q % 8 == 3 && q / 8 < 4) ld_a(q / 8, it);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = an[m];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = bn[i];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int DS, int GL, int IL>
static void run(const char* name, float* out, unsigned long long* clk, const float* rnd) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iter = 10000;
  for (int blocks : {256, 512, 768}) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k_rate<DS, GL, IL>), dim3(blocks), dim3(256), 0, 0, out, clk, iter, rnd);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[2 * 1024]; hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, tick = 0; for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; tick += h[2 * i + 1]; }
    const double wps = blocks / 256.0;
    printf("%-44s %d wave/SIMD: %.2f ns per MFMA per SIMD, clock %.3f GHz, %.1f TFLOP/s\n", name, (int)wps, ms * 1e6 / (iter * 32.0 * wps), cyc / tick / 10.0,
           blocks * 4.0 * iter * 32.0 * 2048 / (ms * 1e-3) / 1e12);
  }
}
int main() {
  float *out, *rnd; unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * 256 * 4096); hipMalloc(&clk, sizeof(unsigned long long) * 2 * 4096);
  const int NR = 8192 + 256 + 260 * 256;
  float* hr = (float*)malloc(sizeof(float) * NR);
  srand(1); for (int i = 0; i < NR; ++i) hr[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMalloc(&rnd, sizeof(float) * NR); hipMemcpy(rnd, hr, sizeof(float) * NR, hipMemcpyHostToDevice);
  run<0, 0, 0>("registers only", out, clk, rnd);
  run<1, 0, 0>("+ 4 ds_read_b128 ahead", out, clk, rnd);
  run<0, 1, 0>("+ 8 global_load_dword ahead", out, clk, rnd);
  run<0, 2, 0>("+ 2 global_load_dwordx4 ahead", out, clk, rnd);
  run<1, 1, 0>("+ 4 ds_read_b128 + 8 dword ahead", out, clk, rnd);
  run<1, 2, 0>("+ 4 ds_read_b128 + 2 dwordx4 ahead", out, clk, rnd);
  run<1, 1, 1>("+ 4 ds_read_b128 + 8 dword interleaved", out, clk, rnd);
  run<1, 2, 1>("+ 4 ds_read_b128 + 2 dwordx4 interleaved", out, clk, rnd);
  return 0;
}
