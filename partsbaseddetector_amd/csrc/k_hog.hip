// k_hog.hip — fused HOG: gradient + orientation snap + bilinear binning +
// block energy + 4-way normalisation + 32-feature emit, one workgroup per
// TC x TC tile of output cells of one pyramid level, all levels in one launch.
// Reference: HOGFeatures<T>::features<uint8_t>, src/HOGFeatures.cpp:168-341.
//
// Bit-exactness plan (compiled with -ffp-contract=off):
//  * the reference SCATTERS each pixel into <=4 block histograms in raster
//    order (:262-265); every histogram bin is an independent accumulator of type T,
//    so a GATHER in which one thread owns one block's bins and adds its contributing
//    pixels in the same raster order produces the same bits, with no atomics;
//  * weights are (wy*wx)*|g| exactly as (:262-265) evaluate them (products of T
//    commute);
//  * the orientation snap (:243-249) is a pure function of the winning channel's integer
//    differences (dx, dy) in [-255, 255]^2: it is TABULATED once per handle by k_hog_binlut,
//    which runs the reference's own chain of T multiply-adds and comparisons for all 511 x 511
//    pairs (one table per T: the dot products are evaluated in T), and looked up per pixel;
//  * the four normalisers are evaluated in double like the reference's
//    `1.0f / sqrt(float_sum + eps)` (:293-299), texture gains in double (:331).
// The chip-wide bound of this kernel is vector-instruction issue (SQ counters: 25 M wave-instructions per 640x480
// frame before round 4, 60 % of the issue slots of its run time), so the phases are organised around instruction count:
// wide staging loads, the tabulated snap, one thread per BLOCK for the histograms and one thread per CELL for the 32
// features (the 18 contrast-sensitive features, the texture sums and the normalisers are shared work inside a cell).
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

// ---- orientation snap table ------------------------------------------------------------------------------------
// lut[(dy + 255) * 511 + (dx + 255)] = best_o of src/HOGFeatures.cpp:243-249 for the gradient (dx, dy), evaluated in T
#define HOG_LUT_SIDE 511
template <typename T>
__device__ __forceinline__ int hog_snap(T dx, T dy) {
  const T uu[9] = {1.000, 0.9397, 0.7660, 0.5000, 0.1736, -0.1736, -0.5000, -0.7660, -0.9397};
  const T vv[9] = {0.000, 0.3420, 0.6428, 0.8660, 0.9848, 0.9848, 0.8660, 0.6428, 0.3420};
  T best_dot = 0;
  int best_o = 0;
#pragma unroll
  for (int o = 0; o < 9; ++o) {
    T dot = uu[o] * dx + vv[o] * dy;
    if (dot > best_dot) { best_dot = dot; best_o = o; }
    else if (-dot > best_dot) { best_dot = -dot; best_o = o + 9; }
  }
  return best_o;
}
template <typename T>
__global__ __launch_bounds__(256) void k_hog_binlut(uint8_t* __restrict__ lut) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HOG_LUT_SIDE * HOG_LUT_SIDE) return;
  const int iy = i / HOG_LUT_SIDE, ix = i - iy * HOG_LUT_SIDE;
  lut[i] = (uint8_t)hog_snap<T>((T)(ix - 255), (T)(iy - 255));
}
size_t hog_binlut_bytes() { return (size_t)HOG_LUT_SIDE * HOG_LUT_SIDE; }
void launch_hog_binlut(uint8_t* lut, int ts, hipStream_t s) {
  const int n = HOG_LUT_SIDE * HOG_LUT_SIDE;
  if (ts == 8) hipLaunchKernelGGL(k_hog_binlut<double>, dim3((n + 255) / 256), dim3(256), 0, s, lut);
  else hipLaunchKernelGGL(k_hog_binlut<float>, dim3((n + 255) / 256), dim3(256), 0, s, lut);
}

struct HogLds {
  int PT;        // pixel window side
  int NB;        // blocks per side (TC+2)
  int P0;        // window rows / columns kept before the first pixel that contributes to the tile's first block
  int MG;        // raw-tile margin before the window (source clamping can reach back sbin/2 pixels)
  int RT;        // raw tile side (pixels)
  int RP;        // raw tile row pitch in bytes (multiple of 4: rows are staged with 4-byte loads)
  int QS;        // (|g|, bin) planes: a row holds its pixels de-interleaved by the cell size — pixel wx at (wx % sbin) * QS + wx / sbin — so
                 // that the histogram walk's lanes (one block each: sbin pixels apart) read CONSECUTIVE words instead of words sbin apart
                 // (4-way bank conflicts on every read with 4-pixel cells; round 6)
  int MP;        // (|g|, bin) plane row pitch in elements = sbin * QS >= PT
  size_t mag_off, bin_off, hist_off, norm_off, ninv_off, tab_off, raw_off, out_off, total;
};

// need_ip: the kernel reads the per-row / per-column block indices (ipy, ipx) — the generic walk of odd or run-time cell sizes; the compile-time even-cell instantiations do not,
// and without the two tables the benched tile (4-pixel cells, 16 x 16 cells, 8-bit BGR) asks for 53 424 B instead of 54 032: THREE workgroups fit a CU's 160 KB instead of two
// (round 6: the hardware allocates LDS in 1 280-byte granules — 43 of them x 3 = 165 120 B did not fit; SQ counters had shown ~11 resident wavefronts per CU, not 18)
__host__ __device__ inline HogLds hog_lds_layout(int sbin, int tc, int bpp, int ts, bool need_ip = true) {   // ts = sizeof(T); bpp = bytes per pixel (channels x element size)
  HogLds L;
  L.NB = tc + 2;
  // A pixel y feeds the blocks floor((y + 0.5) / sbin - 0.5) and the next one (:252-255): block b receives exactly the
  // 2*sbin pixels from b*sbin - sbin/2 on when sbin is even, so NB blocks need (NB + 1) * sbin window rows.  Odd cell
  // sizes keep one spare row on either side.
  L.P0 = (sbin & 1) ? 1 : 0;
  L.PT = L.NB * sbin + sbin + 2 * L.P0;
  L.MG = sbin / 2 + 2;
  L.RT = L.PT + L.MG + 1;
  L.RP = (bpp & 7) ? (L.RT * bpp + 3) & ~3 : (L.RT * bpp + 7) & ~7;   // (8-byte pixels elements: rows stay 8-byte aligned)
  L.QS = (L.PT + sbin - 1) / sbin;
  L.MP = L.QS * sbin;
  size_t o = 0;
  // (|g|, bin) per window pixel are dead once the histograms are complete: the block energies, the normalisers and the
  // staging area of the finished features (half a tile of cells at a time) are written over them (barriers separate the phases)
  const size_t nn = (size_t)ts * (L.NB * L.NB + (tc + 1) * (tc + 1));
  L.mag_off = o; L.norm_off = o; L.ninv_off = o + (size_t)ts * L.NB * L.NB;
  L.out_off = (o + nn + 15) & ~(size_t)15;
  {
    const size_t a = (size_t)ts * L.PT * L.MP, b = L.out_off + (size_t)ts * ((tc * tc + 1) / 2) * (PBD_FLEN + 1);
    o += ((a > b ? a : b) + 15) & ~(size_t)15;
  }
  // the staged source pixels (raw) are dead once (|g|, bin) are computed and the histograms are not live
  // before: they share one region (one barrier more)
  const size_t hist_bytes = (size_t)ts * L.NB * L.NB * PBD_NORIENT, raw_bytes = (size_t)L.RT * L.RP;
  L.hist_off = o; L.raw_off = o;
  o += ((hist_bytes > raw_bytes ? hist_bytes : raw_bytes) + 15) & ~(size_t)15;
  L.tab_off = o; o += ((size_t)ts * 2 + (need_ip ? sizeof(int) : 0)) * 2 * L.PT;  // w0, w1 (, ip) for y and x
  L.bin_off = (o + 3) & ~(size_t)3; o = L.bin_off + (size_t)L.PT * L.MP;
  L.total = (o + 15) & ~(size_t)15;
  return L;
}
size_t hog_lds_bytes(int sbin, int tc, int ts, int bpp) { return hog_lds_layout(sbin, tc, bpp, ts).total; }   // (the planner's bound: with the index tables)

#define HOG_NT 384   // threads per workgroup: (TC+2)^2 = 324 block histograms finish in one pass

// SBIN_T / TC_T > 0: compile-time cell size / tile side (index divisions become shifts, loops unroll);
// 0: taken from the runtime arguments (generic fallback).
// IT: pixel type of the level images (src/HOGFeatures.cpp:136-146: features<uint8_t | uint16_t | float | double>); everything but the
// staging (bytes) and the gradient phase is the same code
template <typename T, int SBIN_T, int TC_T, typename IT = uint8_t>
__global__ __launch_bounds__(HOG_NT, 5) void k_hog(const HogTile* __restrict__ tiles, const LevelDev* __restrict__ levels,
                                                const uint8_t* __restrict__ pyr, T* __restrict__ feat, int cn,
                                                int sbin_rt, int tc_rt, const uint8_t* __restrict__ binlut, uint16_t* __restrict__ split, int split_parts) {
  const int sbin = SBIN_T > 0 ? SBIN_T : sbin_rt;
  const int tc = TC_T > 0 ? TC_T : tc_rt;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PBD_HOG_PRIO     // experiment builds only
  __builtin_amdgcn_s_setprio(PBD_HOG_PRIO);
#endif
  HOG_STAMP(0);
  const HogTile t = tiles[blockIdx.x];
  const LevelDev lv = levels[t.level];
  const int bpp = cn * (int)sizeof(IT);
  constexpr bool NEED_IP = !(SBIN_T > 0 && (SBIN_T & 1) == 0);   // (the even compile-time cell sizes walk their blocks' footprints without the index tables)
  const HogLds L = hog_lds_layout(sbin, tc, bpp, (int)sizeof(T), NEED_IP);
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
  const int PT = L.PT, NB = L.NB, RT = L.RT, RP = L.RP, tid = threadIdx.x;
  const int w = lv.iw, h = lv.ih, bw = lv.bw, bh = lv.bh;
  const int vw = bw * sbin, vh = bh * sbin;  // :176 visible
  const uint8_t* im = pyr + lv.img_off;
  const int stride = w * bpp;
  // pixel window origin: first pixel that can touch block (cy0, cx0) (one more before it for odd cell sizes)
  const int py0 = t.cy0 * sbin - (sbin + 1) / 2 - L.P0, px0 = t.cx0 * sbin - (sbin + 1) / 2 - L.P0;
  const int ry0 = py0 - L.MG, rx0 = px0 - L.MG;  // raw tile origin (source coordinates)

  // ---- stage the source pixels of the window (+margins) in LDS: row r of the raw tile = the RT * cn source bytes from
  //      pixel (ry0 + r, rx0) on, fetched as 4-byte words wherever the word lies inside the source row (the level images
  //      are tightly packed byte rows: no alignment to rely on, the hardware takes unaligned dword loads); the few words
  //      that straddle the row's ends are fetched byte by byte.  Rows / bytes outside the image are never READ by the
  //      gradient pass (it clamps to [1, w - 2] x [1, h - 2], :208,218): rows are clamped to keep the addresses valid,
  //      bytes outside the row are left as they are. ----
  {
    const int wpr = RP >> 2;                          // words per raw row
    const int nword = RT * wpr;
    const int rowbytes = stride;                      // bytes of a source row
    const int b00 = rx0 * bpp;                        // source byte of the raw row's first byte (may be negative)
    // Every word comes from ONE unaligned 4-byte load at an address clamped into the source row, and a word that straddles an end of the row is shifted
    // into place: its bytes outside the row hold whatever the shift brings in — never read (above).  No branch per word: round 6 found the first form
    // (`inside the row ? one load : four byte loads`) compiled to a wait behind EVERY load — a block's 13 words per thread were 13 memory round trips in a
    // row, a third of the block's time.  Positions past the tile's last word repeat it (the same value stored again).
    constexpr int LBW = 7;                            // words in flight per thread and batch (16 x 16-cell tiles of 8-bit BGR: 13 words per thread, two batches)
    if (rowbytes >= 4) {
      // (word i -> (raw row, word of the row) by a multiply-high: i * wpr < 2^32; the LDS address and the shift are worked out again behind the loads instead
      //  of waiting in registers: the kernel shares its SIMDs with the other batches' distance transforms and filter bank, whose wavefronts need the registers —
      //  with 60 instead of 32 registers per lane the stage was 6 % faster and the three-batch pipeline 1 % slower, session 32)
      const unsigned wmagic = 0xFFFFFFFFu / (unsigned)wpr + 1u;
      auto place = [&](int i, int& b0, int& la) {
        const int r = wpr > 1 ? (int)__umulhi((unsigned)i, wmagic) : i, k = i - r * wpr;
        b0 = b00 + 4 * k;
        la = r * RP + 4 * k;
        return min(max(ry0 + r, 0), h - 1);
      };
      for (int i0 = tid; i0 < nword; i0 += HOG_NT * LBW) {
        unsigned wv[LBW];
#pragma unroll
        for (int j = 0; j < LBW; ++j) {
          int b0, la;
          const int sy = place(min(i0 + j * HOG_NT, nword - 1), b0, la);
          wv[j] = *(const __attribute__((aligned(1))) unsigned*)(im + (size_t)sy * stride + min(max(b0, 0), rowbytes - 4));
        }
#pragma unroll
        for (int j = 0; j < LBW; ++j) {
          int b0, la;
          place(min(i0 + j * HOG_NT, nword - 1), b0, la);
          const int sh = min(max(b0, 0), rowbytes - 4) - b0;   // > 0: the word begins in front of the row; < 0: it ends behind it; |sh| >= 4: no byte of it lies in the row
          const int a8 = 8 * min(abs(sh), 3);
          *(unsigned*)(raw + la) = sh > 0 ? wv[j] << a8 : wv[j] >> a8;
        }
      }
    } else {                                          // a source row shorter than a word (1-3 bytes): byte by byte
      for (int i = tid; i < nword; i += HOG_NT) {
        const int r = i / wpr, k = i - r * wpr;
        const int sy = min(max(ry0 + r, 0), h - 1);
        const int b0 = b00 + 4 * k;
        const uint8_t* src = im + (size_t)sy * stride;
        unsigned v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) v |= (unsigned)src[min(max(b0 + q, 0), rowbytes - 1)] << (8 * q);
        *(unsigned*)(raw + r * RP + 4 * k) = v;
      }
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
    if (isx) { wx0[j] = v0; wx1[j] = v1; if constexpr (NEED_IP) ipx[j] = ip; }
    else { wy0[j] = v0; wy1[j] = v1; if constexpr (NEED_IP) ipy[j] = ip; }
  }
  __syncthreads();
  HOG_STAMP(1);

  // ---- per-pixel gradient magnitude + orientation bin (:202-249) ----
  // (One pixel per thread and iteration.  Round 4 measured four unpredicated pixels in flight: 0.0543 against 0.0496 ms per frame; round 6 again, with the channel count at
  //  compile time so that the pixels' LDS reads and table look-ups really go out together — 2 / 4 pixels: 0.0497 / 0.0495 against 0.0464 ms in batches, 0.0795 / 0.0790
  //  against 0.0767 alone, profiles/r06/r06_session39_*: the phase is not waiting for one pixel's chain, and the unpredicated form works on the halo's invisible pixels.)
  for (int i = tid; i < PT * PT; i += HOG_NT) {
    const int wy = i / PT, wx = i - wy * PT;
    const int y = py0 + wy, x = px0 + wx;
    // outside the visible range (:202-203 loops over 1 <= x, y <= visible - 2): |g| = +0 in bin 0 — a histogram bin starts at +0
    // and only ever receives non-negative terms, and x + (+0) == x bit for bit for every x >= +0, so the pixel needs no
    // predicate in the histogram walk
    T m = (T)0;
    int b = 0;
    if (y >= 1 && y < vh - 1 && x >= 1 && x < vw - 1) {
      const int sx = min(x, w - 2), sy = min(y, h - 2);  // :208,218 source clamp
      if constexpr (sizeof(IT) > 1) {
        // wider pixels: the differences in IT's promoted type (int / float / double), squares and comparisons in T, the snap by the
        // reference's own chain (:205-249; no table: the differences are not 9-bit integers)
        const IT* s = (const IT*)(raw + (sy - ry0) * RP + (sx - rx0) * bpp);
        const int rs = RP / (int)sizeof(IT);
        T dx, dy, v;
        if (cn == 1) {
          dy = (T)(s[rs] - s[-rs]);
          dx = (T)(s[1] - s[-1]);
          v = dx * dx + dy * dy;
        } else {
          const T dyb = (T)(s[rs] - s[-rs]), dxb = (T)(s[3] - s[-3]);
          const T vb = dxb * dxb + dyb * dyb;
          const T dyg = (T)(s[rs + 1] - s[-rs + 1]), dxg = (T)(s[4] - s[-2]);
          const T vg = dxg * dxg + dyg * dyg;
          dy = (T)(s[rs + 2] - s[-rs + 2]); dx = (T)(s[5] - s[-1]);
          v = dx * dx + dy * dy;
          if (vg > v) { v = vg; dx = dxg; dy = dyg; }
          if (vb > v) { v = vb; dx = dxb; dy = dyb; }
        }
        b = hog_snap<T>(dx, dy);
        m = t_sqrt(v);
      } else {
      const uint8_t* s = raw + (sy - ry0) * RP + (sx - rx0) * cn;
      int dxi, dyi, vi;
      if (cn == 1) {
        dyi = (int)s[RP] - (int)s[-RP];
        dxi = (int)s[1] - (int)s[-1];
      } else {
        // the squared norms are integers <= 2 * 255^2: exact in T, so the reference's comparisons of T values (:238-239) are
        // comparisons of these integers
        const int dyb = (int)s[RP] - (int)s[-RP], dxb = (int)s[3] - (int)s[-3];
        const int dyg = (int)s[RP + 1] - (int)s[-RP + 1], dxg = (int)s[4] - (int)s[-2];
        dyi = (int)s[RP + 2] - (int)s[-RP + 2]; dxi = (int)s[5] - (int)s[-1];
        vi = dxi * dxi + dyi * dyi;
        const int vg = dxg * dxg + dyg * dyg, vb = dxb * dxb + dyb * dyb;
        if (vg > vi) { vi = vg; dxi = dxg; dyi = dyg; }
        if (vb > vi) { vi = vb; dxi = dxb; dyi = dyb; }
      }
      vi = dxi * dxi + dyi * dyi;
      b = binlut[(dyi + 255) * HOG_LUT_SIDE + (dxi + 255)];
      m = t_sqrt((T)vi);
      }
    }
    const int wq = wx / sbin, pi = wy * L.MP + (wx - wq * sbin) * L.QS + wq;   // de-interleaved by the cell size (HogLds::QS)
    mag[pi] = m;
    bin[pi] = (uint8_t)b;
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
    // hist is BIN-major — bin o of block bl at hist[o * NB * NB + bl] (rounds 1-5: [bl][18], whose 18-word pitch put lanes l and l + 16
    // on one bank whatever their bins): with equal bins the lanes of a group sit on consecutive banks, and the energy / feature passes
    // below read consecutive words
    T* hb = hist + bl;
    const int NBB = NB * NB;
    if constexpr (SBIN_T > 0 && (SBIN_T & 1) == 0) {
      // even cell size: block b receives exactly the 2 * sbin window rows / columns from b * sbin on, the first sbin of them
      // with the "upper" weight (the pixel's iy is b - 1: weight vy0), the rest with the "lower" one (iy == b: vy1)
      const int wy_lo = lby * SBIN_T, wx_lo = lbx * SBIN_T;
      T fxv[2 * SBIN_T];
#pragma unroll
      for (int dx = 0; dx < 2 * SBIN_T; ++dx) fxv[dx] = dx < SBIN_T ? wx0[wx_lo + dx] : wx1[wx_lo + dx];
#pragma unroll 2
      for (int dy = 0; dy < 2 * SBIN_T; ++dy) {
        const int wy = wy_lo + dy;
        const T fy = dy < SBIN_T ? wy0[wy] : wy1[wy];
        const uint8_t* brow = bin + wy * L.MP + lbx;      // pixel wx_lo + dx of the row: (dx % sbin) * QS + lbx + dx / sbin
        const T* mrow = mag + wy * L.MP + lbx;
#pragma unroll
        for (int dx = 0; dx < 2 * SBIN_T; ++dx) {
          const int pi = (dx % SBIN_T) * L.QS + dx / SBIN_T;
          hb[brow[pi] * NBB] += (fy * fxv[dx]) * mrow[pi];
        }
      }
    } else {
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
          const int wq = wx / sbin, pi = wy * L.MP + (wx - wq * sbin) * L.QS + wq;
          hb[bin[pi] * NBB] += (fy * fx) * mag[pi];
        }
      }
    }
  }
  __syncthreads();
  HOG_STAMP(3);

  // ---- block energy (:270-283) ----
  for (int i = tid; i < NB * NB; i += HOG_NT) {
    const T* hsrc = hist + i;
    const int NBB = NB * NB;
    T acc = (T)0;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
      T s = hsrc[o * NBB] + hsrc[(o + 9) * NBB];
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
  // ---- 32 features per cell (:301-338), ONE thread per cell: the 18 products val * n_i are computed once and feed the
  //      contrast-sensitive features AND the four texture sums (sequential adds in orientation order, as the reference's
  //      loop accumulates t1..t4); a lane-per-feature mapping recomputed them and ran every wavefront through all three
  //      feature kinds.  Half a tile of cells at a time: the finished features are staged in LDS (row of 33 per cell) and
  //      written out with 32 consecutive lanes per cell (128 B, coalesced). ----
  T* out = feat + lv.cell_off * PBD_FLEN;
  T* stg = (T*)(smem + L.out_off);
  const int ncell = tc * tc, half = (ncell + 1) / 2;
  for (int c0 = 0; c0 < ncell; c0 += half) {
    const int cell = c0 + tid;
    if (tid < half && cell < ncell) {
      const int ly = cell / tc, lx = cell - ly * tc;
      if (t.cy0 + ly < lv.ch && t.cx0 + lx < lv.cw) {
        const T n1 = ninv[(ly + 1) * NC + lx + 1], n2 = ninv[ly * NC + lx + 1];
        const T n3 = ninv[(ly + 1) * NC + lx], n4 = ninv[ly * NC + lx];
        const T* hsrc = hist + ((ly + 1) * NB + lx + 1);
        const int NBB = NB * NB;
        T* d = stg + tid * (PBD_FLEN + 1);
        T t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#pragma unroll 6
        for (int o = 0; o < PBD_NORIENT; ++o) {                    // :305-317
          const T val = hsrc[o * NBB];
          const T h1 = t_fmin(val * n1, (T)0.2), h2 = t_fmin(val * n2, (T)0.2);
          const T h3 = t_fmin(val * n3, (T)0.2), h4 = t_fmin(val * n4, (T)0.2);
          d[o] = (T)(0.5 * (double)(h1 + h2 + h3 + h4));
          t1 += h1; t2 += h2; t3 += h3; t4 += h4;
        }
#pragma unroll 3
        for (int o = 0; o < PBD_NORIENT / 2; ++o) {                // :320-328
          const T sum = hsrc[o * NBB] + hsrc[(o + PBD_NORIENT / 2) * NBB];
          const T h1 = t_fmin(sum * n1, (T)0.2), h2 = t_fmin(sum * n2, (T)0.2);
          const T h3 = t_fmin(sum * n3, (T)0.2), h4 = t_fmin(sum * n4, (T)0.2);
          d[PBD_NORIENT + o] = (T)(0.5 * (double)(h1 + h2 + h3 + h4));
        }
        d[27] = (T)(0.2357 * (double)t1);                          // :331-334
        d[28] = (T)(0.2357 * (double)t2);
        d[29] = (T)(0.2357 * (double)t3);
        d[30] = (T)(0.2357 * (double)t4);
        d[31] = (T)0;                                              // :337
      }
    }
    __syncthreads();
    for (int i = tid; i < half * PBD_FLEN; i += HOG_NT) {
      const int k = i & 31, lc = i >> 5;
      const int cl = c0 + lc;
      const int ly = cl / tc, lx = cl - ly * tc;
      const int cy = t.cy0 + ly, cx = t.cx0 + lx;
      if (cl < ncell && cy < lv.ch && cx < lv.cw) {
        const T v = stg[lc * (PBD_FLEN + 1) + k];
        const size_t gc = (size_t)cy * lv.cw + cx;
        out[gc * PBD_FLEN + k] = v;
        if constexpr (sizeof(T) == 4) {
          // PBD_CONV_SPLIT: the feature's three exact bfloat16 parts for the split-product filter bank, [cell][split][32] (k_conv_split.hip:
          // v = h + m + l, every subtraction exact) — written here instead of by a separate pass over the features
          if (split && split_parts == 3) {
            uint16_t* sp = split + (lv.cell_off + gc) * (3 * PBD_FLEN) + k;
            float r = v;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const unsigned u = __float_as_uint(r);
              const unsigned hb = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
              sp[q * PBD_FLEN] = (uint16_t)hb;
              r = r - __uint_as_float(hb << 16);
            }
          } else if (split) {   // PBD_CONV_SPLIT_F16: two binary16 parts of v 2^12 (k_feat_split16)
            _Float16* sp = (_Float16*)split + (lv.cell_off + gc) * (2 * PBD_FLEN) + k;
            const float x = v * 4096.f;
            const _Float16 hv = (_Float16)x;
            sp[0] = hv;
            sp[PBD_FLEN] = (_Float16)(x - (float)hv);
          }
        }
      }
    }
    __syncthreads();
  }
  HOG_STAMP(5);
}

template <typename T>
static void launch_hog_t(const HogTile* tiles, int ntiles, const LevelDev* levels, const uint8_t* pyr, T* feat,
                         int cn, int sbin, int tc, const uint8_t* binlut, uint16_t* split, int split_parts, int depth, hipStream_t s) {
  const int esz = depth == PBD_DEPTH_16U ? 2 : depth == PBD_DEPTH_32F ? 4 : depth == PBD_DEPTH_64F ? 8 : 1;
  auto go = [&](auto kern, bool need_ip) {
    const size_t lds = hog_lds_layout(sbin, tc, cn * esz, (int)sizeof(T), need_ip).total;   // (the kernel's own layout: NEED_IP)
    static LdsOptIn optin;  // one per instantiation (the lambda is instantiated per kernel), per-device state inside
    optin.ensure((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(HOG_NT), lds, s, tiles, levels, pyr, feat, cn, sbin, tc, binlut, split, split_parts);
  };
  if (depth == PBD_DEPTH_16U) go(k_hog<T, 0, 0, uint16_t>, true);      // (the generic cell size / tile side instantiation: pbd_detect_image's depths)
  else if (depth == PBD_DEPTH_32F) go(k_hog<T, 0, 0, float>, true);
  else if (depth == PBD_DEPTH_64F) go(k_hog<T, 0, 0, double>, true);
  else if (sbin == 4 && tc == 16) go(k_hog<T, 4, 16>, false);
  else if (sbin == 4 && tc == 8) go(k_hog<T, 4, 8>, false);
  else if (sbin == 8 && tc == 8) go(k_hog<T, 8, 8>, false);
  else go(k_hog<T, 0, 0>, true);
}

// ts = sizeof(T) of the handle's instantiation (HOGFeatures<float> / HOGFeatures<double>, src/HOGFeatures.cpp:51-52);
// binlut: the orientation-snap table of the same T (launch_hog_binlut)
// split != nullptr (float handles with a split-product filter bank): the features' parts are written too — split_parts 3: bfloat16, 2: binary16
void launch_hog(const HogTile* tiles, int ntiles, const LevelDev* levels, const uint8_t* pyr, void* feat, int ts,
                int cn, int sbin, int tc, const uint8_t* binlut, uint16_t* split, int split_parts, int depth, hipStream_t s) {
  if (ntiles <= 0) return;
  if (ts == 8) launch_hog_t<double>(tiles, ntiles, levels, pyr, (double*)feat, cn, sbin, tc, binlut, nullptr, 0, depth, s);
  else launch_hog_t<float>(tiles, ntiles, levels, pyr, (float*)feat, cn, sbin, tc, binlut, split, split_parts, depth, s);
}
