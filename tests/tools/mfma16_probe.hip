// Ad-hoc probe (not part of the product): result layout of v_mfma_f32_16x16x4_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_layout(float* out) {
  const int l = threadIdx.x;
  const float a = 1000.f * (l & 15) + (l >> 4);      // hypothesis A[i = l&15][k = l>>4] = 1000 i + k
  f32x4 acc = {0, 0, 0, 0};
  const float b = ((l >> 4) == 2) ? 1.f : 0.f;       // B[k][j] = (k == 2) -> D[i][j] = 1000 i + 2
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
  f32x4 acc2 = {0, 0, 0, 0};
  const float b2 = ((l >> 4) == 0) ? (float)(l & 15) : 0.f;   // D[i][j] = j
  acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, b2, acc2, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = acc2[r];
}
int main() {
  float* d; hipMalloc(&d, 512 * sizeof(float));
  k_layout<<<1, 64>>>(d);
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) printf("lane %2d: %6.0f %6.0f %6.0f %6.0f | %3.0f %3.0f %3.0f %3.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3],
                                                       h[256 + l * 4], h[256 + l * 4 + 1], h[256 + l * 4 + 2], h[256 + l * 4 + 3]);
  return 0;
}
