// Ad-hoc probe: resident blocks per CU as the runtime computes them, against block size, LDS and VGPR count.
// hipcc --offload-arch=gfx950 -O3 tests/tools/occ_probe.hip -o /tmp/occ_probe && /tmp/occ_probe
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ char smem[];
template <int V>
__global__ void kern(float* out) {
  float r[V];
#pragma unroll
  for (int i = 0; i < V; ++i) r[i] = out[threadIdx.x + i * 64];
  __syncthreads();
  float s = 0;
#pragma unroll
  for (int i = 0; i < V; ++i) s += r[i] * r[(i * 7 + 3) % V];
  out[threadIdx.x] = s + smem[threadIdx.x];
}
template <int V>
static void probe() {
  for (int nt : {64, 128, 256})
    for (size_t lds : {(size_t)1024, (size_t)20480}) {
      hipFuncSetAttribute((const void*)kern<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      int nb = -1;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern<V>, nt, lds);
      hipFuncAttributes a;
      hipFuncGetAttributes(&a, (const void*)kern<V>);
      printf("V %3d numRegs %3d nt %3d lds %6zu -> %d blocks/CU = %d waves/CU\n", V, a.numRegs, nt, lds, nb, nb * nt / 64);
    }
}
int main() {
  probe<8>(); probe<60>(); probe<90>(); probe<120>();
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("regsPerBlock %d regsPerMultiprocessor %d maxThreadsPerMultiProcessor %d sharedMemPerMultiprocessor %zu maxBlocksPerMultiProcessor %d\n",
         p.regsPerBlock, p.regsPerMultiprocessor, p.maxThreadsPerMultiProcessor, p.sharedMemPerMultiprocessor, p.maxBlocksPerMultiProcessor);
  return 0;
}
