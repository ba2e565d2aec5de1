// k_hog.hip — fused HOG: gradient + orientation snap + bilinear binning +
// block energy + 4-way normalisation + 32-feature emit, one workgroup per
// TC x TC tile of output cells of one pyramid level, all levels in one launch.
// Reference: HOGFeatures<T>::features<uint8_t>, src/HOGFeatures.cpp:168-341.
//
// Bit-exactness plan (compiled with -ffp-contract=off):
//  * the reference SCATTERS each pixel into <=4 block histograms in raster
//    order (:262-265); every histogram bin is an independent float
//    accumulator, so a GATHER in which one thread owns one (block, orientation)
//    bin and adds its contributing pixels in the same raster order produces
//    the same bits, with no atomics;
//  * weights are (wy*wx)*|g| exactly as (:262-265) evaluate them (float
//    products commute);
//  * the four normalisers are evaluated in double like the reference's
//    `1.0f / sqrt(float_sum + eps)` (:293-299), texture gains in double (:331).
// HBM-bound and small (25 MB algorithmic per 640x480 frame): pixels are read
// through L2 (each level image is a few hundred KB), per-pixel (|g|, bin) and
// the tile's histograms live in LDS, the output is written cell-major with 32
// consecutive lanes covering one cell's 128 B.
#include "pbd_internal.hpp"

// debug: per-phase wall-clock stamps (100 MHz) of block 0 of the last k_hog launch
__device__ unsigned long long pbd_hog_dbg[8];
#define HOG_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) pbd_hog_dbg[i] = wall_clock64(); } while (0)
void hog_debug_read(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(pbd_hog_dbg), sizeof(unsigned long long) * 8); }

struct HogLds {
  int PT;        // pixel window side
  int NB;        // blocks per side (TC+2)
  int MG;        // raw-tile margin before the window (source clamping can reach back sbin/2 pixels)
  int RT;        // raw tile side
  size_t mag_off, bin_off, hist_off, norm_off, ninv_off, tab_off, raw_off, total;
};

__host__ __device__ inline HogLds hog_lds_layout(int sbin, int tc, int cn) {
  HogLds L;
  L.NB = tc + 2;
  L.PT = L.NB * sbin + sbin + 2;
  L.MG = sbin / 2 + 2;
  L.RT = L.PT + L.MG + 1;
  size_t o = 0;
  L.mag_off = o; o += sizeof(float) * L.PT * L.PT;
  L.hist_off = o; o += sizeof(float) * L.NB * L.NB * PBD_NORIENT;
  L.norm_off = o; o += sizeof(float) * L.NB * L.NB;
  L.ninv_off = o; o += sizeof(float) * (tc + 1) * (tc + 1);
  L.tab_off = o; o += (sizeof(float) * 2 + sizeof(int)) * 2 * L.PT;  // w0,w1,ip for y and x
  L.bin_off = o; o += L.PT * L.PT;
  o = (o + 3) & ~(size_t)3;
  L.raw_off = o; o += (size_t)L.RT * L.RT * cn;
  L.total = (o + 15) & ~(size_t)15;
  return L;
}
size_t hog_lds_bytes(int sbin, int tc) { return hog_lds_layout(sbin, tc, 3).total; }

#define HOG_NT 384   // threads per workgroup: (TC+2)^2 = 324 block histograms finish in one pass

// SBIN_T / TC_T > 0: compile-time cell size / tile side (index divisions become shifts, loops unroll);
// 0: taken from the runtime arguments (generic fallback).
template <int SBIN_T, int TC_T>
__global__ __launch_bounds__(HOG_NT) void k_hog(const HogTile* __restrict__ tiles, const LevelDev* __restrict__ levels,
                                                const uint8_t* __restrict__ pyr, float* __restrict__ feat, int cn,
                                                int sbin_rt, int tc_rt) {
  const int sbin = SBIN_T > 0 ? SBIN_T : sbin_rt;
  const int tc = TC_T > 0 ? TC_T : tc_rt;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  HOG_STAMP(0);
  const HogTile t = tiles[blockIdx.x];
  const LevelDev lv = levels[t.level];
  const HogLds L = hog_lds_layout(sbin, tc, 3);
  float* mag = (float*)(smem + L.mag_off);
  uint8_t* bin = (uint8_t*)(smem + L.bin_off);
  uint8_t* raw = (uint8_t*)(smem + L.raw_off);
  float* hist = (float*)(smem + L.hist_off);
  float* norm = (float*)(smem + L.norm_off);
  float* ninv = (float*)(smem + L.ninv_off);
  float* wy0 = (float*)(smem + L.tab_off);  // vy0 per window row
  float* wy1 = wy0 + L.PT;
  float* wx0 = wy1 + L.PT;
  float* wx1 = wx0 + L.PT;
  int* ipy = (int*)(wx1 + L.PT);
  int* ipx = ipy + L.PT;
  const int PT = L.PT, NB = L.NB, RT = L.RT, tid = threadIdx.x;
  const int w = lv.iw, h = lv.ih, bw = lv.bw, bh = lv.bh;
  const int vw = bw * sbin, vh = bh * sbin;  // :176 visible
  const uint8_t* im = pyr + lv.img_off;
  const int stride = w * cn;
  // pixel window origin: first pixel that can touch block (cy0, cx0), minus one for safety
  const int py0 = t.cy0 * sbin - sbin / 2 - 1, px0 = t.cx0 * sbin - sbin / 2 - 1;
  const int ry0 = py0 - L.MG, rx0 = px0 - L.MG;  // raw tile origin (source coordinates, clamped on load)

  // ---- stage the source pixels of the window (+margins) in LDS, coalesced byte rows ----
  const int rowb = RT * cn;
  for (int cb = tid; cb < rowb; cb += HOG_NT) {  // one byte column per thread, 32 independent row loads in flight
    const int xc = cb / cn, chn = cb - xc * cn;
    const int sx = min(max(rx0 + xc, 0), w - 1);
    const uint8_t* col = im + sx * cn + chn;
    for (int r0 = 0; r0 < RT; r0 += 32) {
      uint8_t rr[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const int sy = min(max(ry0 + min(r0 + j, RT - 1), 0), h - 1);
        rr[j] = col[(size_t)sy * stride];
      }
#pragma unroll
      for (int j = 0; j < 32; ++j)
        if (r0 + j < RT) raw[(r0 + j) * rowb + cb] = rr[j];
    }
  }
  // ---- interpolation tables per window row / column (:252-260) ----
  for (int i = tid; i < 2 * PT; i += HOG_NT) {
    const bool isx = i >= PT;
    const int j = isx ? i - PT : i;
    const int p = (isx ? px0 : py0) + j;
    float pp = (float)(((double)(float)p + 0.5) / (double)(float)sbin - 0.5);
    int ip = (int)floorf(pp);
    float v0 = pp - (float)ip;
    float v1 = (float)(1.0 - (double)v0);
    if (isx) { wx0[j] = v0; wx1[j] = v1; ipx[j] = ip; }
    else { wy0[j] = v0; wy1[j] = v1; ipy[j] = ip; }
  }
  __syncthreads();
  HOG_STAMP(1);

  // ---- per-pixel gradient magnitude + orientation bin (:202-249) ----
  const float uu[9] = {1.000, 0.9397, 0.7660, 0.5000, 0.1736, -0.1736, -0.5000, -0.7660, -0.9397};
  const float vv[9] = {0.000, 0.3420, 0.6428, 0.8660, 0.9848, 0.9848, 0.8660, 0.6428, 0.3420};
  for (int i = tid; i < PT * PT; i += HOG_NT) {
    const int wy = i / PT, wx = i - wy * PT;
    const int y = py0 + wy, x = px0 + wx;
    float m = 0.f;
    int b = 255;
    if (y >= 1 && y < vh - 1 && x >= 1 && x < vw - 1) {
      const int sx = min(x, w - 2), sy = min(y, h - 2);  // :208,218 source clamp
      const uint8_t* s = raw + ((sy - ry0) * RT + (sx - rx0)) * cn;
      float dx, dy, v;
      if (cn == 1) {
        dy = (float)((int)s[rowb] - (int)s[-rowb]);
        dx = (float)((int)s[1] - (int)s[-1]);
        v = dx * dx + dy * dy;
      } else {
        float dyb = (float)((int)s[rowb] - (int)s[-rowb]);
        float dxb = (float)((int)s[3] - (int)s[-3]);
        float vb = dxb * dxb + dyb * dyb;
        float dyg = (float)((int)s[rowb + 1] - (int)s[-rowb + 1]);
        float dxg = (float)((int)s[4] - (int)s[-2]);
        float vg = dxg * dxg + dyg * dyg;
        dy = (float)((int)s[rowb + 2] - (int)s[-rowb + 2]);
        dx = (float)((int)s[5] - (int)s[-1]);
        v = dx * dx + dy * dy;
        if (vg > v) { v = vg; dx = dxg; dy = dyg; }
        if (vb > v) { v = vb; dx = dxb; dy = dyb; }
      }
      float best_dot = 0;
      int best_o = 0;
#pragma unroll
      for (int o = 0; o < 9; ++o) {
        float dot = uu[o] * dx + vv[o] * dy;
        if (dot > best_dot) { best_dot = dot; best_o = o; }
        else if (-dot > best_dot) { best_dot = -dot; best_o = o + 9; }
      }
      m = sqrtf(v);
      b = best_o;
    }
    mag[i] = m;
    bin[i] = (uint8_t)b;
  }
  for (int i = tid; i < NB * NB * PBD_NORIENT; i += HOG_NT) hist[i] = 0.f;
  __syncthreads();
  HOG_STAMP(2);

  // ---- histogram: one thread owns one block's 18 bins and walks the block's pixels in the
  //      reference's raster order (:262-265), so every bin sees the same sequence of float adds ----
  for (int bl = tid; bl < NB * NB; bl += HOG_NT) {
    const int lby = bl / NB, lbx = bl - lby * NB;
    const int by = t.cy0 + lby, bx = t.cx0 + lbx;
    if (by >= bh || bx >= bw) continue;
    float* hb = hist + bl * PBD_NORIENT;
    const int wy_lo = lby * sbin, wx_lo = lbx * sbin, span = 2 * sbin + 2;
#pragma unroll 2
    for (int dy = 0; dy < span; ++dy) {
      const int wy = wy_lo + dy;
      if (wy >= PT) break;
      const int iy = ipy[wy];
      float fy;
      if (iy == by) fy = wy1[wy]; else if (iy == by - 1) fy = wy0[wy]; else continue;
      for (int dx = 0; dx < span; ++dx) {
        const int wx = wx_lo + dx;
        if (wx >= PT) break;
        const int ix = ipx[wx];
        float fx;
        if (ix == bx) fx = wx1[wx]; else if (ix == bx - 1) fx = wx0[wx]; else continue;
        const int o = bin[wy * PT + wx];
        if (o == 255) continue;
        hb[o] += (fy * fx) * mag[wy * PT + wx];
      }
    }
  }
  __syncthreads();
  HOG_STAMP(3);

  // ---- block energy (:270-283) ----
  for (int i = tid; i < NB * NB; i += HOG_NT) {
    const float* hsrc = hist + i * PBD_NORIENT;
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
      float s = hsrc[o] + hsrc[o + 9];
      acc += s * s;
    }
    norm[i] = acc;
  }
  __syncthreads();

  // ---- normalisers on the (TC+1)^2 block corners (:292-299) ----
  const int NC = tc + 1;
  for (int i = tid; i < NC * NC; i += HOG_NT) {
    const int y = i / NC, x = i - y * NC;
    const float* p = norm + y * NB + x;
    float s = p[0] + p[1] + p[NB] + p[NB + 1];
    ninv[i] = (float)(1.0f / sqrt((double)s + 0.0001));
  }
  __syncthreads();

  HOG_STAMP(4);
  // ---- 32 features per cell, one lane per feature (:301-338) ----
  float* out = feat + lv.cell_off * PBD_FLEN;
  for (int i = tid; i < tc * tc * PBD_FLEN; i += HOG_NT) {
    const int k = i & 31;
    const int cell = i >> 5;
    const int ly = cell / tc, lx = cell - ly * tc;
    const int cy = t.cy0 + ly, cx = t.cx0 + lx;
    if (cy >= lv.ch || cx >= lv.cw) continue;
    const float n1 = ninv[(ly + 1) * NC + lx + 1], n2 = ninv[ly * NC + lx + 1];
    const float n3 = ninv[(ly + 1) * NC + lx], n4 = ninv[ly * NC + lx];
    const float* hsrc = hist + ((ly + 1) * NB + lx + 1) * PBD_NORIENT;
    float r;
    if (k < 27) {
      float val = (k < 18) ? hsrc[k] : hsrc[k - 18] + hsrc[k - 9];
      float h1 = fminf(val * n1, 0.2f), h2 = fminf(val * n2, 0.2f);
      float h3 = fminf(val * n3, 0.2f), h4 = fminf(val * n4, 0.2f);
      r = (float)(0.5 * (double)(h1 + h2 + h3 + h4));
    } else if (k < 31) {
      const float n = (k == 27) ? n1 : (k == 28) ? n2 : (k == 29) ? n3 : n4;
      float tsum = 0.f;
#pragma unroll
      for (int o = 0; o < PBD_NORIENT; ++o) tsum += fminf(hsrc[o] * n, 0.2f);
      r = (float)(0.2357 * (double)tsum);
    } else {
      r = 0.f;
    }
    out[((size_t)cy * lv.cw + cx) * PBD_FLEN + k] = r;
  }
  HOG_STAMP(5);
}

void launch_hog(const HogTile* tiles, int ntiles, const LevelDev* levels, const uint8_t* pyr, float* feat,
                int cn, int sbin, int tc, hipStream_t s) {
  if (ntiles <= 0) return;
  const size_t lds = hog_lds_bytes(sbin, tc);
  auto go = [&](auto kern) {
    static size_t configured = 0;  // one per instantiation
    if (lds > configured) {
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      configured = lds;
    }
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(HOG_NT), lds, s, tiles, levels, pyr, feat, cn, sbin, tc);
  };
  if (sbin == 4 && tc == 16) go(k_hog<4, 16>);
  else if (sbin == 8 && tc == 8) go(k_hog<8, 8>);
  else go(k_hog<0, 0>);
}
