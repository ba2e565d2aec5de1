// k_pyramid.hip — 8-bit image pyramid of HOGFeatures<T>::pyramid
// (reference src/HOGFeatures.cpp:111-127): cv::resize(INTER_LINEAR) for the
// `interval` levels of the first octave and cv::pyrDown chains below them.
// Pure integer arithmetic (OpenCV 2.4 fixed-point definitions);
// bit-exact against oracle/pbd_oracle.c.  Small (7 MB of level images per 640x480 frame) and
// bound by memory-instruction issue, not bytes: the level images are packed byte rows, so the
// kernels move them as 4-byte words wherever a word lies inside a row (the hardware takes
// unaligned dword accesses) and keep the 5x5 pyrDown window in LDS (separable passes).
#include <algorithm>
#include "pbd_internal.hpp"

typedef __attribute__((aligned(1))) unsigned u32_unaligned;

__device__ __forceinline__ int sat_short_dev(float v) {
  int i = __float2int_rn(v);  // cvRound: round half to even
  return i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
}

// One launch for all first-octave levels of all frames of a batch: blockIdx.y = job (frame, level), grid-stride over
// groups of RPX consecutive pixels of an output row (one thread: 4 pixels = 12 bytes of a BGR row, stored as 3 words).
#define RPX 4
__global__ __launch_bounds__(256) void k_resize_linear_u8(const PyrJob* __restrict__ jobs, int cn, int sstride,
                                                          const uint8_t* __restrict__ src0, uint8_t* __restrict__ pyr) {
  const PyrJob a = jobs[blockIdx.y];
  const int dw = a.dw, dh = a.dh, sw = a.sw, sh = a.sh;
  const uint8_t* src = src0 + a.soff;
  uint8_t* dst = pyr + a.doff;
  const bool copy = (dw == sw && dh == sh);
  if (copy) {   // cv::resize to the same size copies (level 0): rows of dw * cn bytes, word by word
    const int rowb = dw * cn, wpr = (rowb + 3) >> 2, nw = wpr * dh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += gridDim.x * blockDim.x) {
      const int y = i / wpr, k = i - y * wpr;
      const uint8_t* s = src + (size_t)y * sstride + 4 * k;
      uint8_t* d = dst + (size_t)y * rowb + 4 * k;
      if (4 * k + 4 <= rowb) *(u32_unaligned*)d = *(const u32_unaligned*)s;
      else for (int b = 0; 4 * k + b < rowb; ++b) d[b] = s[b];
    }
    return;
  }
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  const int gpr = (dw + RPX - 1) / RPX, ngrp = gpr * dh;   // groups per row, groups
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ngrp; i += gridDim.x * blockDim.x) {
    const int dy = i / gpr, gx = i - dy * gpr;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    const int b0 = sat_short_dev((1.f - fy) * 2048), b1 = sat_short_dev(fy * 2048);
    const int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
    const int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
    const uint8_t* R0 = src + (size_t)sy0 * sstride;
    const uint8_t* R1 = src + (size_t)sy1 * sstride;
    uint8_t ob[RPX * 3];
    const int npx = min(RPX, dw - gx * RPX);
#pragma unroll
    for (int p = 0; p < RPX; ++p) {
      const int dx = min(gx * RPX + p, dw - 1);
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = (int)floorf(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      const bool edge = (sx + 1 >= sw);
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
      const int a0 = sat_short_dev((1.f - fx) * 2048), a1 = sat_short_dev(fx * 2048);
      const uint8_t* S0 = R0 + sx * cn;
      const uint8_t* S1 = R1 + sx * cn;
      // the two source pixels of a row are 2 * cn contiguous bytes: for BGR two (unaligned) words per row instead of six byte
      // loads; at the row's last pixel (`edge`: no right neighbour, weights reset) byte loads of the one pixel
      uint8_t p0[6], p1[6];
      if (cn == 3 && !edge) {
        const unsigned w00 = *(const u32_unaligned*)S0, w01 = *(const unsigned short __attribute__((aligned(1)))*)(S0 + 4);
        const unsigned w10 = *(const u32_unaligned*)S1, w11 = *(const unsigned short __attribute__((aligned(1)))*)(S1 + 4);
#pragma unroll
        for (int b = 0; b < 4; ++b) { p0[b] = (uint8_t)(w00 >> (8 * b)); p1[b] = (uint8_t)(w10 >> (8 * b)); }
        p0[4] = (uint8_t)w01; p0[5] = (uint8_t)(w01 >> 8); p1[4] = (uint8_t)w11; p1[5] = (uint8_t)(w11 >> 8);
      } else {
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          const int o = (b < cn || !edge) && b < 2 * cn ? b : 0;
          p0[b] = S0[o]; p1[b] = S1[o];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c < cn) {
          int r0, r1;
          if (edge) { r0 = p0[c] * 2048; r1 = p1[c] * 2048; }
          else { r0 = p0[c] * a0 + p0[c + cn] * a1; r1 = p1[c] * a0 + p1[c + cn] * a1; }
          const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
          ob[p * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      }
    }
    uint8_t* d = dst + ((size_t)dy * dw + gx * RPX) * cn;
    if (cn == 3 && npx == RPX) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        *(u32_unaligned*)(d + 4 * k) = (unsigned)ob[4 * k] | ((unsigned)ob[4 * k + 1] << 8) | ((unsigned)ob[4 * k + 2] << 16) | ((unsigned)ob[4 * k + 3] << 24);
    } else {
      for (int p = 0; p < npx; ++p)
        for (int c = 0; c < cn; ++c) d[p * cn + c] = ob[p * 3 + c];
    }
  }
}

__device__ __forceinline__ int reflect101_dev(int p, int len) {
  if (len == 1) return 0;
  while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// cv::pyrDown: [1 4 6 4 1] x [1 4 6 4 1] / 256 on the 5x5 window centred on (2x, 2y), BORDER_REFLECT_101, (sum + 128) >> 8.
// One launch per octave step: blockIdx.z selects the chain (frame, level j <- level j - interval); a workgroup produces a
// PD_TW x PD_TH tile of the destination from a (2 PD_TW + 3) x (2 PD_TH + 3) source window staged in LDS: horizontal pass
// into 16-bit partial sums, vertical pass out of them (integers: any order of the 25 products gives the same sum).
#define PD_TW 32
#define PD_TH 8
#define PD_SW (2 * PD_TW + 3)
#define PD_SH (2 * PD_TH + 3)
template <int CN>   // channels: compile-time, the index arithmetic below divides by it
__global__ __launch_bounds__(256) void k_pyrdown_u8(const PyrJob* __restrict__ jobs, uint8_t* __restrict__ pyr) {
  constexpr int cn = CN;
  __shared__ __attribute__((aligned(4))) uint8_t tile[PD_SH][(PD_SW * CN + 3) & ~3];
  __shared__ unsigned short hs[PD_SH][PD_TW * CN];
  const PyrJob a = jobs[blockIdx.z];
  const int sw = a.sw, sh = a.sh;
  const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  const int x0 = blockIdx.x * PD_TW, y0 = blockIdx.y * PD_TH;
  if (x0 >= dw || y0 >= dh) return;                     // the grid is sized for the chain's largest level
  const uint8_t* src = pyr + a.soff;
  uint8_t* dst = pyr + a.doff;
  const int tid = threadIdx.x;
  const int sx0 = 2 * x0 - 2, sy0 = 2 * y0 - 2;         // source pixel of tile[0][0]
  const int rowb = PD_SW * cn, srow = sw * cn;
  // ---- stage the source window: words where the word lies inside the source row, reflected bytes at the borders ----
  {
    const int wpr = (rowb + 3) >> 2, nw = wpr * PD_SH;
    const bool xin = sx0 >= 0 && sx0 + PD_SW <= sw;     // the window's columns need no reflection (wave-uniform)
    for (int i = tid; i < nw; i += 256) {
      const int r = i / wpr, k = i - r * wpr;
      const uint8_t* srow_p = src + (size_t)reflect101_dev(sy0 + r, sh) * srow;
      unsigned v;
      if (xin && sx0 * cn + 4 * k + 4 <= srow) {
        v = *(const u32_unaligned*)(srow_p + sx0 * cn + 4 * k);
      } else {
        v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int cb = 4 * k + b;                     // byte of the window row: pixel cb / cn, channel cb % cn
          const int px = cb / cn, ch = cb - px * cn;
          const int sx = reflect101_dev(sx0 + min(px, PD_SW - 1), sw);
          v |= (unsigned)srow_p[sx * cn + ch] << (8 * b);
        }
      }
      *(unsigned*)&tile[r][4 * k] = v;
    }
  }
  __syncthreads();
  // ---- horizontal pass: hs[r][x * cn + c] = sum_j w[j] * tile[r][(2x + j) * cn + c] ----
  {
    const int n = PD_SH * PD_TW * cn;
    for (int i = tid; i < n; i += 256) {
      const int r = i / (PD_TW * cn), xc = i - r * (PD_TW * cn);
      const int x = xc / cn, c = xc - x * cn;
      const uint8_t* p = &tile[r][2 * x * cn + c];
      hs[r][xc] = (unsigned short)(p[0] + 4 * p[cn] + 6 * p[2 * cn] + 4 * p[3 * cn] + p[4 * cn]);
    }
  }
  __syncthreads();
  // ---- vertical pass + store: consecutive lanes = consecutive bytes of a destination row ----
  {
    const int n = PD_TH * PD_TW * cn;
    for (int i = tid; i < n; i += 256) {
      const int y = i / (PD_TW * cn), xc = i - y * (PD_TW * cn);
      const int x = xc / cn;
      if (y0 + y < dh && x0 + x < dw) {
        const int s = hs[2 * y][xc] + 4 * hs[2 * y + 1][xc] + 6 * hs[2 * y + 2][xc] + 4 * hs[2 * y + 3][xc] + hs[2 * y + 4][xc];
        dst[((size_t)(y0 + y) * dw + x0) * cn + xc] = (uint8_t)((s + 128) >> 8);
      }
    }
  }
}

void launch_resize(const PyrJob* jobs, int njobs, int maxpix, int cn, int sstride, const uint8_t* src, uint8_t* pyr, hipStream_t s) {
  if (njobs <= 0) return;
  dim3 grid((maxpix / RPX + 255) / 256 + 1, njobs);
  hipLaunchKernelGGL(k_resize_linear_u8, grid, dim3(256), 0, s, jobs, cn, sstride, src, pyr);
}

// maxw / maxh: the largest DESTINATION level of the launch's chains
void launch_pyrdown(const PyrJob* jobs, int njobs, int maxw, int maxh, int cn, uint8_t* pyr, hipStream_t s) {
  if (njobs <= 0) return;
  dim3 grid((maxw + PD_TW - 1) / PD_TW, (maxh + PD_TH - 1) / PD_TH, njobs);
  if (cn == 3) hipLaunchKernelGGL(k_pyrdown_u8<3>, grid, dim3(256), 0, s, jobs, pyr);
  else hipLaunchKernelGGL(k_pyrdown_u8<1>, grid, dim3(256), 0, s, jobs, pyr);
}

// ---------------------------------------------------------------------------------------------------------------------
// The other image depths HOGFeatures<T>::pyramid accepts (src/HOGFeatures.cpp:136-146: CV_16U, CV_32F, CV_64F;
// pbd_detect_image).  No caller of the reference passes them (src/demo.cpp, ros/Node.cpp, cells/detect.cpp hand over
// 8-bit BGR), so these kernels are plain — one thread per output element — and follow oracle/pbd_oracle.c's
// restatement of OpenCV 2.4 operation by operation (-ffp-contract=off):
//   resize:  floating-point interpolation with float coefficients — row value S[sx] a0 + S[sx + cn] a1 in WT, then
//            R0 b0 + R1 b1 in WT (WT = float for ushort / float pixels, double for double), ushort: cvRound + clamp;
//   pyrDown: ushort = the 8-bit integer form; float / double: row = s2 6 + (s1 + s3) 4 + s0 + s4 per source row, the
//            same expression over the five rows, times 1 / 256 (FltCast<T, 8>, scalar association).
// Job offsets are in BYTES, strides in elements.
// ---------------------------------------------------------------------------------------------------------------------
template <typename PT> struct PyrWork { typedef float type; };
template <> struct PyrWork<double> { typedef double type; };
template <typename PT> __device__ __forceinline__ PT pyr_cast(typename PyrWork<PT>::type v) { return (PT)v; }
template <> __device__ __forceinline__ uint16_t pyr_cast<uint16_t>(float v) {
  const int i = __float2int_rn(v);   // cvRound: half to even
  return (uint16_t)(i < 0 ? 0 : (i > 65535 ? 65535 : i));
}

template <typename PT>
__global__ __launch_bounds__(256) void k_resize_linear_any(const PyrJob* __restrict__ jobs, int cn, int sstride,
                                                           const uint8_t* __restrict__ src0, uint8_t* __restrict__ pyr) {
  typedef typename PyrWork<PT>::type WT;
  const PyrJob a = jobs[blockIdx.y];
  const int dw = a.dw, dh = a.dh, sw = a.sw, sh = a.sh;
  const PT* src = (const PT*)(src0 + a.soff);
  PT* dst = (PT*)(pyr + a.doff);
  const int n = dw * dh * cn;
  if (dw == sw && dh == sh) {   // cv::resize to the same size: all fractions are zero — a copy
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const int y = i / (dw * cn), xc = i - y * (dw * cn);
      dst[i] = src[(size_t)y * sstride + xc];
    }
    return;
  }
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int dy = i / (dw * cn), xc = i - dy * (dw * cn);
    const int dx = xc / cn, c = xc - dx * cn;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    const bool edge = (sx + 1 >= sw);
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const float a0 = 1.f - fx, a1 = fx;
    const WT b0 = 1.f - fy, b1 = fy;
    const int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
    const int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
    const PT* S0 = src + (size_t)sy0 * sstride + sx * cn + c;
    const PT* S1 = src + (size_t)sy1 * sstride + sx * cn + c;
    WT r0, r1;
    if (edge) { r0 = (WT)(S0[0] * 1); r1 = (WT)(S1[0] * 1); }
    else { r0 = (WT)(S0[0] * a0 + S0[cn] * a1); r1 = (WT)(S1[0] * a0 + S1[cn] * a1); }
    dst[i] = pyr_cast<PT>(r0 * b0 + r1 * b1);
  }
}

template <typename PT>
__global__ __launch_bounds__(256) void k_pyrdown_any(const PyrJob* __restrict__ jobs, int cn, uint8_t* __restrict__ pyr) {
  const PyrJob a = jobs[blockIdx.y];
  const int sw = a.sw, sh = a.sh, dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  const PT* src = (const PT*)(pyr + a.soff);
  PT* dst = (PT*)(pyr + a.doff);
  const int n = dw * dh * cn;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int y = i / (dw * cn), xc = i - y * (dw * cn);
    const int x = xc / cn, c = xc - x * cn;
    int xs[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) xs[j] = reflect101_dev(2 * x + j - 2, sw) * cn + c;
    if constexpr (sizeof(PT) == 2) {
      int row[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const PT* S = src + (size_t)reflect101_dev(2 * y + k - 2, sh) * sw * cn;
        row[k] = (int)S[xs[0]] + 4 * (int)S[xs[1]] + 6 * (int)S[xs[2]] + 4 * (int)S[xs[3]] + (int)S[xs[4]];
      }
      dst[i] = (PT)((row[0] + 4 * row[1] + 6 * row[2] + 4 * row[3] + row[4] + 128) >> 8);
    } else {
      PT row[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const PT* S = src + (size_t)reflect101_dev(2 * y + k - 2, sh) * sw * cn;
        const PT s0 = S[xs[0]], s1 = S[xs[1]], s2 = S[xs[2]], s3 = S[xs[3]], s4 = S[xs[4]];
        row[k] = s2 * 6 + (s1 + s3) * 4 + s0 + s4;
      }
      dst[i] = (row[2] * 6 + (row[1] + row[3]) * 4 + row[0] + row[4]) * (PT)(1. / 256);
    }
  }
}

// depth: PBD_DEPTH_16U / 32F / 64F; sstride: bytes
void launch_resize_any(const PyrJob* jobs, int njobs, int maxpix, int cn, int depth, int sstride, const uint8_t* src, uint8_t* pyr, hipStream_t s) {
  if (njobs <= 0) return;
  dim3 grid((unsigned)std::min<long long>(((long long)maxpix * cn + 255) / 256, 4096), njobs);
  if (depth == PBD_DEPTH_16U) hipLaunchKernelGGL(k_resize_linear_any<uint16_t>, grid, dim3(256), 0, s, jobs, cn, sstride / 2, src, pyr);
  else if (depth == PBD_DEPTH_32F) hipLaunchKernelGGL(k_resize_linear_any<float>, grid, dim3(256), 0, s, jobs, cn, sstride / 4, src, pyr);
  else hipLaunchKernelGGL(k_resize_linear_any<double>, grid, dim3(256), 0, s, jobs, cn, sstride / 8, src, pyr);
}
void launch_pyrdown_any(const PyrJob* jobs, int njobs, int maxpix, int cn, int depth, uint8_t* pyr, hipStream_t s) {
  if (njobs <= 0) return;
  dim3 grid((unsigned)std::min<long long>(((long long)maxpix * cn + 255) / 256, 4096), njobs);
  if (depth == PBD_DEPTH_16U) hipLaunchKernelGGL(k_pyrdown_any<uint16_t>, grid, dim3(256), 0, s, jobs, cn, pyr);
  else if (depth == PBD_DEPTH_32F) hipLaunchKernelGGL(k_pyrdown_any<float>, grid, dim3(256), 0, s, jobs, cn, pyr);
  else hipLaunchKernelGGL(k_pyrdown_any<double>, grid, dim3(256), 0, s, jobs, cn, pyr);
}
