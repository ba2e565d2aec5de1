// k_pyramid.hip — 8-bit image pyramid of HOGFeatures<T>::pyramid
// (reference src/HOGFeatures.cpp:111-127): cv::resize(INTER_LINEAR) for the
// `interval` levels of the first octave and cv::pyrDown chains below them.
// Pure integer arithmetic (OpenCV 2.4 fixed-point definitions);
// bit-exact against oracle/pbd_oracle.c.  HBM/L2-bound, tiny: one thread per
// output pixel, all channels; interleaved reads stay within a few cache lines.
#include "pbd_internal.hpp"

__device__ __forceinline__ int sat_short_dev(float v) {
  int i = __float2int_rn(v);  // cvRound: round half to even
  return i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
}

// One launch for all first-octave levels of all frames of a batch: blockIdx.y = job (frame, level), grid-stride over pixels.
__global__ __launch_bounds__(256) void k_resize_linear_u8(const PyrJob* __restrict__ jobs, int cn, int sstride,
                                                          const uint8_t* __restrict__ src0, uint8_t* __restrict__ pyr) {
  const PyrJob a = jobs[blockIdx.y];
  const int dw = a.dw, dh = a.dh, sw = a.sw, sh = a.sh;
  const uint8_t* src = src0 + a.soff;
  uint8_t* dst = pyr + a.doff;
  const int npix = dw * dh;
  const bool copy = (dw == sw && dh == sh);
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    const int dy = i / dw, dx = i - dy * dw;
    if (copy) {
      for (int c = 0; c < cn; ++c) dst[(size_t)i * cn + c] = src[(size_t)dy * sstride + dx * cn + c];
      continue;
    }
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    const bool edge = (sx + 1 >= sw);
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const int a0 = sat_short_dev((1.f - fx) * 2048), a1 = sat_short_dev(fx * 2048);
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    const int b0 = sat_short_dev((1.f - fy) * 2048), b1 = sat_short_dev(fy * 2048);
    int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
    int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
    const uint8_t* S0 = src + (size_t)sy0 * sstride + sx * cn;
    const uint8_t* S1 = src + (size_t)sy1 * sstride + sx * cn;
    for (int c = 0; c < cn; ++c) {
      int r0, r1;
      if (edge) { r0 = S0[c] * 2048; r1 = S1[c] * 2048; }
      else { r0 = S0[c] * a0 + S0[c + cn] * a1; r1 = S1[c] * a0 + S1[c + cn] * a1; }
      int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
      dst[(size_t)i * cn + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

__device__ __forceinline__ int reflect101_dev(int p, int len) {
  if (len == 1) return 0;
  while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// One launch per octave step: blockIdx.y selects the chain (frame, level j <- level j-interval).
__global__ __launch_bounds__(256) void k_pyrdown_u8(const PyrJob* __restrict__ jobs, int cn, uint8_t* __restrict__ pyr) {
  const PyrJob a = jobs[blockIdx.y];
  const int sw = a.sw, sh = a.sh;
  const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  const uint8_t* src = pyr + a.soff;
  uint8_t* dst = pyr + a.doff;
  const int wt[5] = {1, 4, 6, 4, 1};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dw * dh; i += gridDim.x * blockDim.x) {
    const int y = i / dw, x = i - y * dw;
    int sxs[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) sxs[j] = reflect101_dev(2 * x + j - 2, sw) * cn;
    int sum[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const uint8_t* row = src + (size_t)reflect101_dev(2 * y + r - 2, sh) * sw * cn;
      for (int c = 0; c < cn; ++c) {
        int rs = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) rs += wt[j] * row[sxs[j] + c];
        sum[c] += wt[r] * rs;
      }
    }
    for (int c = 0; c < cn; ++c) dst[(size_t)i * cn + c] = (uint8_t)((sum[c] + 128) >> 8);
  }
}

void launch_resize(const PyrJob* jobs, int njobs, int maxpix, int cn, int sstride, const uint8_t* src, uint8_t* pyr, hipStream_t s) {
  if (njobs <= 0) return;
  dim3 grid((maxpix + 255) / 256, njobs);
  hipLaunchKernelGGL(k_resize_linear_u8, grid, dim3(256), 0, s, jobs, cn, sstride, src, pyr);
}

void launch_pyrdown(const PyrJob* jobs, int njobs, int maxpix, int cn, uint8_t* pyr, hipStream_t s) {
  if (njobs <= 0) return;
  dim3 grid((maxpix + 255) / 256, njobs);
  hipLaunchKernelGGL(k_pyrdown_u8, grid, dim3(256), 0, s, jobs, cn, pyr);
}
