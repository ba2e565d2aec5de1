// Ad-hoc probe (not part of the product): operand/result layout and issue rate of v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
// D = A(16x4) * B(4x16): feed A[i][k] = 100*i + k encoded by lane guesses, B = one-hot to read A back, etc.
__global__ void k_layout(double* out) {
  const int l = threadIdx.x;
  // hypothesis: A lane l holds A[i = l&15][k = l>>4]; B lane l holds B[k = l>>4][j = l&15]
  const double a = 1000.0 * (l & 15) + (l >> 4);      // A[i][k] = 1000 i + k
  f64x4 acc = {0, 0, 0, 0};
  // B = selector: B[k][j] = (k == 2) ? 1 : 0  -> D[i][j] = A[i][2] = 1000 i + 2 for all j
  const double b = ((l >> 4) == 2) ? 1.0 : 0.0;
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
  // second: B[k][j] = (k == 0) ? j : 0, A[i][k] = 1 -> D[i][j] = j
  f64x4 acc2 = {0, 0, 0, 0};
  const double b2 = ((l >> 4) == 0) ? (double)(l & 15) : 0.0;
  acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, b2, acc2, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = acc2[r];
}
template <int NACC>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters, double a, double b) {
  f64x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a + u, b + n, acc[n], 0, 0, 0);
  }
  double s = 0;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern, int bpc, int nacc) {
  double* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(double));
  const int iters = 4000, grid = 256 * bpc;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 256>>>(d, 10, 1.0, 2.0);
  hipEventRecord(e0);
  kern<<<grid, 256>>>(d, iters, 1.0, 2.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)grid * 4 * iters * 8 * nacc;
  printf("%-24s blocks/CU=%d %.3f ms %.1f TFLOP/s (%.1f cycles/MFMA/SIMD @2.2GHz)\n", name, bpc, ms, n * 2048 / ms / 1e9, ms * 1e-3 * 2.2e9 / (n / 1024.0));
  hipFree(d);
}
int main() {
  double* d; hipMalloc(&d, 512 * sizeof(double));
  k_layout<<<1, 64>>>(d);
  double h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("test1 (expect D[i][j] = 1000 i + 2): lane: regs\n");
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) printf("  lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  printf("test2 (expect D[i][j] = j):\n");
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) printf("  lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[256 + l * 4], h[256 + l * 4 + 1], h[256 + l * 4 + 2], h[256 + l * 4 + 3]);
  for (int b : {1, 3}) { run("16x16x4 f64, 1 acc", k_rate<1>, b, 1); run("16x16x4 f64, 2 acc", k_rate<2>, b, 2); run("16x16x4 f64, 4 acc", k_rate<4>, b, 4); }
  return 0;
}
