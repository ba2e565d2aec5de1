// rcp_probe.hip — accuracy of v_rcp_f64 + one Newton step against the IEEE reciprocal on gfx950, over the
// denominators the distance transform divides by (den = 2a*dx, a a float deformation weight, dx < 2048).
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tests/tools/rcp_probe.hip -o tests/tools/rcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void k(const double* den, int n, unsigned long long* hist0, unsigned long long* hist1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = den[i];
  const double ref = 1.0 / d;
  const double x0 = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, x0, 1.0);
  const double x1 = __builtin_fma(x0, e, x0);
  long long a = __double_as_longlong(ref), b0 = __double_as_longlong(x0), b1 = __double_as_longlong(x1);
  long long u0 = llabs(a - b0), u1 = llabs(a - b1);
  atomicAdd(&hist0[u0 > 63 ? 63 : u0], 1ull);
  atomicAdd(&hist1[u1 > 63 ? 63 : u1], 1ull);
}
int main() {
  const int n = 1 << 24;
  double* h = (double*)malloc(sizeof(double) * n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    float a = 0.0005f + (float)rand() / RAND_MAX * (i % 3 ? 0.06f : 1.2f);
    int dx = 1 + rand() % 2047;
    h[i] = (2.0 * -(double)a) * (double)dx;
  }
  double* d; unsigned long long *h0, *h1, r0[64], r1[64];
  hipMalloc(&d, sizeof(double) * n); hipMalloc(&h0, 512); hipMalloc(&h1, 512);
  hipMemcpy(d, h, sizeof(double) * n, hipMemcpyHostToDevice); hipMemset(h0, 0, 512); hipMemset(h1, 0, 512);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n, h0, h1);
  hipMemcpy(r0, h0, 512, hipMemcpyDeviceToHost); hipMemcpy(r1, h1, 512, hipMemcpyDeviceToHost);
  printf("ulp distance from RN(1/den): v_rcp_f64 alone / + one Newton step\n");
  for (int u = 0; u < 64; ++u) if (r0[u] || r1[u]) printf("  %2d%s ulp: %llu / %llu\n", u, u == 63 ? "+" : "", r0[u], r1[u]);
  return 0;
}
