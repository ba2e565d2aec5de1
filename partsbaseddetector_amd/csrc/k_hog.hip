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
#ifdef PBD_PROBES
__device__ unsigned long long pbd_hog_dbg[8];
#define HOG_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) pbd_hog_dbg[i] = wall_clock64(); } while (0)
void hog_debug_read(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(pbd_hog_dbg), sizeof(unsigned long long) * 8); }
#else
#define HOG_STAMP(i) do { } while (0)
void hog_debug_read(unsigned long long* out) { for (int i = 0; i < 8; ++i) out[i] = 0; }
#endif

struct HogLds {
  int PT;        // pixel window side
  int NB;        // blocks per side (TC+2)
  int P0;        // window rows / columns kept before the first pixel that contributes to the tile's first block
  int MG;        // raw-tile margin before the window (source clamping can reach back sbin/2 pixels)
  int RT;        // raw tile side
  size_t mag_off, bin_off, hist_off, norm_off, ninv_off, tab_off, raw_off, total;
};

__host__ __device__ inline HogLds hog_lds_layout(int sbin, int tc, int cn, int ts) {   // ts = sizeof(T)
  HogLds L;
  L.NB = tc + 2;
  // A pixel y feeds the blocks floor((y + 0.5) / sbin - 0.5) and the next one (:252-255): block b receives exactly the
  // 2*sbin pixels from b*sbin - sbin/2 on when sbin is even, so NB blocks need (NB + 1) * sbin window rows.  Odd cell
  // sizes keep one spare row on either side.  (For sbin 4 / 16-cell tiles: 76 instead of 78 rows — with the
  // overlays below 52.8 KB of LDS per workgroup, i.e. THREE workgroups per CU: the 604 tiles of a 640x480 pyramid
  // are then resident at once instead of in two generations.)
  L.P0 = (sbin & 1) ? 1 : 0;
  L.PT = L.NB * sbin + sbin + 2 * L.P0;
  L.MG = sbin / 2 + 2;
  L.RT = L.PT + L.MG + 1;
  size_t o = 0;
  // (|g|, bin) per window pixel are dead once the histograms are complete: the block energies and the normalisers
  // are written over them (a barrier separates the phases)
  L.mag_off = o; L.norm_off = o; L.ninv_off = o + (size_t)ts * L.NB * L.NB;
  {
    const size_t a = (size_t)ts * L.PT * L.PT, b = (size_t)ts * (L.NB * L.NB + (tc + 1) * (tc + 1));
    o += ((a > b ? a : b) + 7) & ~(size_t)7;
  }
  // the staged source pixels (raw) are dead once (|g|, bin) are computed and the histograms are not live
  // before: they share one region (one barrier more)
  const size_t hist_bytes = (size_t)ts * L.NB * L.NB * PBD_NORIENT, raw_bytes = (size_t)L.RT * L.RT * cn;
  L.hist_off = o; L.raw_off = o;
  o += ((hist_bytes > raw_bytes ? hist_bytes : raw_bytes) + 7) & ~(size_t)7;
  L.tab_off = o; o += ((size_t)ts * 2 + sizeof(int)) * 2 * L.PT;  // w0,w1,ip for y and x
  L.bin_off = o; o += L.PT * L.PT;
  L.total = (o + 15) & ~(size_t)15;
  return L;
}
size_t hog_lds_bytes(int sbin, int tc, int ts) { return hog_lds_layout(sbin, tc, 3, ts).total; }

#define HOG_NT 384   // threads per workgroup: (TC+2)^2 = 324 block histograms finish in one pass

// SBIN_T / TC_T > 0: compile-time cell size / tile side (index divisions become shifts, loops unroll);
// 0: taken from the runtime arguments (generic fallback).
template <typename T, int SBIN_T, int TC_T>
__global__ __launch_bounds__(HOG_NT) void k_hog(const HogTile* __restrict__ tiles, const LevelDev* __restrict__ levels,
                                                const uint8_t* __restrict__ pyr, T* __restrict__ feat, int cn,
                                                int sbin_rt, int tc_rt) {
  const int sbin = SBIN_T > 0 ? SBIN_T : sbin_rt;
  const int tc = TC_T > 0 ? TC_T : tc_rt;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  HOG_STAMP(0);
  const HogTile t = tiles[blockIdx.x];
  const LevelDev lv = levels[t.level];
  const HogLds L = hog_lds_layout(sbin, tc, 3, (int)sizeof(T));
  T* mag = (T*)(smem + L.mag_off);
  uint8_t* bin = (uint8_t*)(smem + L.bin_off);
  uint8_t* raw = (uint8_t*)(smem + L.raw_off);
  T* hist = (T*)(smem + L.hist_off);
  T* norm = (T*)(smem + L.norm_off);
  T* ninv = (T*)(smem + L.ninv_off);
  T* wy0 = (T*)(smem + L.tab_off);  // vy0 per window row
  T* wy1 = wy0 + L.PT;
  T* wx0 = wy1 + L.PT;
  T* wx1 = wx0 + L.PT;
  int* ipy = (int*)(wx1 + L.PT);
  int* ipx = ipy + L.PT;
  const int PT = L.PT, NB = L.NB, RT = L.RT, tid = threadIdx.x;
  const int w = lv.iw, h = lv.ih, bw = lv.bw, bh = lv.bh;
  const int vw = bw * sbin, vh = bh * sbin;  // :176 visible
  const uint8_t* im = pyr + lv.img_off;
  const int stride = w * cn;
  // pixel window origin: first pixel that can touch block (cy0, cx0) (one more before it for odd cell sizes)
  const int py0 = t.cy0 * sbin - (sbin + 1) / 2 - L.P0, px0 = t.cx0 * sbin - (sbin + 1) / 2 - L.P0;
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
    T pp = (T)(((double)(T)p + 0.5) / (double)(T)sbin - 0.5);
    int ip = (int)t_floor(pp);
    T v0 = pp - (T)ip;
    T v1 = (T)(1.0 - (double)v0);
    if (isx) { wx0[j] = v0; wx1[j] = v1; ipx[j] = ip; }
    else { wy0[j] = v0; wy1[j] = v1; ipy[j] = ip; }
  }
  __syncthreads();
  HOG_STAMP(1);

  // ---- per-pixel gradient magnitude + orientation bin (:202-249) ----
  const T uu[9] = {1.000, 0.9397, 0.7660, 0.5000, 0.1736, -0.1736, -0.5000, -0.7660, -0.9397};
  const T vv[9] = {0.000, 0.3420, 0.6428, 0.8660, 0.9848, 0.9848, 0.8660, 0.6428, 0.3420};
  for (int i = tid; i < PT * PT; i += HOG_NT) {
    const int wy = i / PT, wx = i - wy * PT;
    const int y = py0 + wy, x = px0 + wx;
    T m = (T)0;
    int b = 255;
    if (y >= 1 && y < vh - 1 && x >= 1 && x < vw - 1) {
      const int sx = min(x, w - 2), sy = min(y, h - 2);  // :208,218 source clamp
      const uint8_t* s = raw + ((sy - ry0) * RT + (sx - rx0)) * cn;
      T dx, dy, v;
      if (cn == 1) {
        dy = (T)((int)s[rowb] - (int)s[-rowb]);
        dx = (T)((int)s[1] - (int)s[-1]);
        v = dx * dx + dy * dy;
      } else {
        T dyb = (T)((int)s[rowb] - (int)s[-rowb]);
        T dxb = (T)((int)s[3] - (int)s[-3]);
        T vb = dxb * dxb + dyb * dyb;
        T dyg = (T)((int)s[rowb + 1] - (int)s[-rowb + 1]);
        T dxg = (T)((int)s[4] - (int)s[-2]);
        T vg = dxg * dxg + dyg * dyg;
        dy = (T)((int)s[rowb + 2] - (int)s[-rowb + 2]);
        dx = (T)((int)s[5] - (int)s[-1]);
        v = dx * dx + dy * dy;
        if (vg > v) { v = vg; dx = dxg; dy = dyg; }
        if (vb > v) { v = vb; dx = dxb; dy = dyb; }
      }
      T best_dot = 0;
      int best_o = 0;
#pragma unroll
      for (int o = 0; o < 9; ++o) {
        T dot = uu[o] * dx + vv[o] * dy;
        if (dot > best_dot) { best_dot = dot; best_o = o; }
        else if (-dot > best_dot) { best_dot = -dot; best_o = o + 9; }
      }
      m = t_sqrt(v);
      b = best_o;
    }
    mag[i] = m;
    bin[i] = (uint8_t)b;
  }
  __syncthreads();                                       // every thread is done with the staged pixels ...
  for (int i = tid; i < NB * NB * PBD_NORIENT; i += HOG_NT) hist[i] = (T)0;   // ... whose LDS the histograms take over
  __syncthreads();
  HOG_STAMP(2);

  // ---- histogram: one thread owns one block's 18 bins and walks the block's pixels in the
  //      reference's raster order (:262-265), so every bin sees the same sequence of T adds ----
  for (int bl = tid; bl < NB * NB; bl += HOG_NT) {
    const int lby = bl / NB, lbx = bl - lby * NB;
    const int by = t.cy0 + lby, bx = t.cx0 + lbx;
    if (by >= bh || bx >= bw) continue;
    T* hb = hist + bl * PBD_NORIENT;
    const int wy_lo = lby * sbin, wx_lo = lbx * sbin, span = 2 * sbin + 2;
#pragma unroll 2
    for (int dy = 0; dy < span; ++dy) {
      const int wy = wy_lo + dy;
      if (wy >= PT) break;
      const int iy = ipy[wy];
      T fy;
      if (iy == by) fy = wy1[wy]; else if (iy == by - 1) fy = wy0[wy]; else continue;
      for (int dx = 0; dx < span; ++dx) {
        const int wx = wx_lo + dx;
        if (wx >= PT) break;
        const int ix = ipx[wx];
        T fx;
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
    const T* hsrc = hist + i * PBD_NORIENT;
    T acc = (T)0;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
      T s = hsrc[o] + hsrc[o + 9];
      acc += s * s;
    }
    norm[i] = acc;
  }
  __syncthreads();

  // ---- normalisers on the (TC+1)^2 block corners (:292-299) ----
  const int NC = tc + 1;
  for (int i = tid; i < NC * NC; i += HOG_NT) {
    const int y = i / NC, x = i - y * NC;
    const T* p = norm + y * NB + x;
    T s = p[0] + p[1] + p[NB] + p[NB + 1];
    ninv[i] = (T)(1.0f / sqrt((double)s + 0.0001));
  }
  __syncthreads();

  HOG_STAMP(4);
  // ---- 32 features per cell, one lane per feature (:301-338) ----
  T* out = feat + lv.cell_off * PBD_FLEN;
  for (int i = tid; i < tc * tc * PBD_FLEN; i += HOG_NT) {
    const int k = i & 31;
    const int cell = i >> 5;
    const int ly = cell / tc, lx = cell - ly * tc;
    const int cy = t.cy0 + ly, cx = t.cx0 + lx;
    if (cy >= lv.ch || cx >= lv.cw) continue;
    const T n1 = ninv[(ly + 1) * NC + lx + 1], n2 = ninv[ly * NC + lx + 1];
    const T n3 = ninv[(ly + 1) * NC + lx], n4 = ninv[ly * NC + lx];
    const T* hsrc = hist + ((ly + 1) * NB + lx + 1) * PBD_NORIENT;
    T r;
    if (k < 27) {
      T val = (k < 18) ? hsrc[k] : hsrc[k - 18] + hsrc[k - 9];
      T h1 = t_fmin(val * n1, (T)0.2), h2 = t_fmin(val * n2, (T)0.2);
      T h3 = t_fmin(val * n3, (T)0.2), h4 = t_fmin(val * n4, (T)0.2);
      r = (T)(0.5 * (double)(h1 + h2 + h3 + h4));
    } else if (k < 31) {
      const T n = (k == 27) ? n1 : (k == 28) ? n2 : (k == 29) ? n3 : n4;
      T tsum = (T)0;
#pragma unroll
      for (int o = 0; o < PBD_NORIENT; ++o) tsum += t_fmin(hsrc[o] * n, (T)0.2);
      r = (T)(0.2357 * (double)tsum);
    } else {
      r = (T)0;
    }
    out[((size_t)cy * lv.cw + cx) * PBD_FLEN + k] = r;
  }
  HOG_STAMP(5);
}

template <typename T>
static void launch_hog_t(const HogTile* tiles, int ntiles, const LevelDev* levels, const uint8_t* pyr, T* feat,
                         int cn, int sbin, int tc, hipStream_t s) {
  const size_t lds = hog_lds_bytes(sbin, tc, (int)sizeof(T));
  auto go = [&](auto kern) {
    static LdsOptIn optin;  // one per instantiation (the lambda is instantiated per kernel), per-device state inside
    optin.ensure((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(HOG_NT), lds, s, tiles, levels, pyr, feat, cn, sbin, tc);
  };
  if (sbin == 4 && tc == 16) go(k_hog<T, 4, 16>);
  else if (sbin == 4 && tc == 8) go(k_hog<T, 4, 8>);
  else if (sbin == 8 && tc == 8) go(k_hog<T, 8, 8>);
  else go(k_hog<T, 0, 0>);
}

// ts = sizeof(T) of the handle's instantiation (HOGFeatures<float> / HOGFeatures<double>, src/HOGFeatures.cpp:51-52)
void launch_hog(const HogTile* tiles, int ntiles, const LevelDev* levels, const uint8_t* pyr, void* feat, int ts,
                int cn, int sbin, int tc, hipStream_t s) {
  if (ntiles <= 0) return;
  if (ts == 8) launch_hog_t<double>(tiles, ntiles, levels, pyr, (double*)feat, cn, sbin, tc, s);
  else launch_hog_t<float>(tiles, ntiles, levels, pyr, (float*)feat, cn, sbin, tc, s);
}
