// k_conv.hip — SpatialConvolutionEngine::pdf as one batched filter-bank
// correlation over all pyramid levels (reference
// src/SpatialConvolutionEngine.cpp:70-124, Filter2D src/filter.cpp:3879-3924).
//
//   resp[l][n](y,x) = sum_c sum_{i,j} w_n[i][j][c] * F_l(y+i-kh/2, x+j-kw/2, c)
//   "same" size, correlation (no flip), constant border: 0 for c<flen-1,
//   1 for the last (truncation) channel (:147-155).
//
// Two kernels behind the same launcher signature:
//  * k_conv_exact — VALU, reproduces the reference's summation order bit for
//    bit: per channel a tap-ordered (row-major) chain of separately rounded
//    mul + add starting from 0 (filter.cpp:3914-3918), then the channel
//    partials are added in channel order (`pdf += pdfc`, :92).  Compiled with
//    -ffp-contract=off so hipcc cannot fuse the mul/add.
//  * k_conv_mfma  — fp32 MFMA implicit GEMM (M = cells, N = filters, K =
//    kh*kw*flen = 800): a k-ordered fma chain, |delta| ~1e-6 vs the reference
//    order; the fast path when nfilters*flen is a real dense contraction.
// Both stage a (T+kh-1)x(T+kw-1)-cell feature tile with halo in LDS once per
// workgroup (border values materialised there) and write plane-major outputs.
#include <algorithm>
#include <vector>
#include <cstring>
#include "pbd_internal.hpp"
#include <type_traits>

// debug: per-phase wall-clock stamps (100 MHz) of one workgroup of the last k_conv_mfma launch
#ifdef PBD_PROBES
__device__ unsigned long long pbd_conv_dbg[8];
#define CONV_STAMP(i) do { if (blockIdx.x == 300 && blockIdx.y == 2 && threadIdx.x == 0) pbd_conv_dbg[i] = wall_clock64(); } while (0)
void conv_debug_read(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(pbd_conv_dbg), sizeof(unsigned long long) * 8); }
// k_conv_glds: per-phase sums over the units of one workgroup (wave 0 lane 0), slots: 0 = barrier waits before the K loops, 1 = both K loops, 2 = barrier before the epilogue, 4 = shader cycles of both K loops (s_memtime), 5 = epilogue; 6 = units; 7 = life
#define GLDS_T(var) const unsigned long long var = wall_clock64()
#define GLDS_C(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define GLDS_ACC(i, a, b) do { if (blockIdx.x == gridDim.x / 2 + 3 && threadIdx.x == 0) pbd_conv_dbg[i] += (b) - (a); } while (0)
#define GLDS_INIT() do { if (blockIdx.x == gridDim.x / 2 + 3 && threadIdx.x == 0) for (int q_ = 0; q_ < 8; ++q_) pbd_conv_dbg[q_] = 0; } while (0)
#else
#define GLDS_T(var) do { } while (0)
#define GLDS_C(var) do { } while (0)
#define GLDS_ACC(i, a, b) do { } while (0)
#define GLDS_INIT() do { } while (0)
#define CONV_STAMP(i) do { } while (0)
void conv_debug_read(unsigned long long* out) { for (int i = 0; i < 8; ++i) out[i] = 0; }
#endif

#define CT 16        // spatial tile side (cells)
#define CSTR 33      // LDS floats per cell (32 + 1 pad: conflict-free across x)
#define NFG 8        // filters held in registers per pass (exact kernel)

// Stage the (CT+KH-1) x (CT+KW-1) cell tile with halo into LDS ([cell][CSTR]).  Eight lanes fetch
// one cell's 128 B as float4s; NB batches of independent loads are in flight before the first wait
// (clamped addresses, the border value is selected after the load: 0, or 1 for the last channel).
template <int KH, int KW>
__device__ __forceinline__ void stage_feature_tile(float* __restrict__ ft, const float* __restrict__ F, int y0, int x0,
                                                   int H, int W, int tid) {
  constexpr int TW = CT + KW - 1, TH = CT + KH - 1, N = TH * TW * 8, NB = (N + 255) / 256;
  float4 r[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int i = min(tid + j * 256, N - 1);
    const int cell = i >> 3, q = i & 7;
    const int ty = cell / TW, tx = cell - ty * TW;
    const int y = min(max(y0 + ty - KH / 2, 0), H - 1), x = min(max(x0 + tx - KW / 2, 0), W - 1);
    r[j] = *(const float4*)(F + ((size_t)y * W + x) * PBD_FLEN + q * 4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int i = tid + j * 256;
    if (i < N) {
      const int cell = i >> 3, q = i & 7;
      const int ty = cell / TW, tx = cell - ty * TW;
      const int y = y0 + ty - KH / 2, x = x0 + tx - KW / 2;
      float4 v = r[j];
      if (!(y >= 0 && y < H && x >= 0 && x < W)) v = make_float4(0.f, 0.f, 0.f, q == 7 ? 1.f : 0.f);
      float* d = ft + cell * CSTR + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
}

// Same staging for any scalar type: 16-byte vectors (float4 / double2), LPC lanes per cell.
template <typename T, int KH, int KW>
__device__ __forceinline__ void stage_feature_tile_t(T* __restrict__ ft, const T* __restrict__ F, int y0, int x0,
                                                     int H, int W, int tid) {
  constexpr int EPV = 16 / (int)sizeof(T), LPC = PBD_FLEN / EPV;
  constexpr int TW = CT + KW - 1, TH = CT + KH - 1, N = TH * TW * LPC, NB = (N + 255) / 256;
  struct alignas(16) V { T e[EPV]; };
  constexpr int BATCH = 8;
  for (int j0 = 0; j0 < NB; j0 += BATCH) {
    V r[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = min(tid + (j0 + j) * 256, N - 1);
      const int cell = i / LPC, q = i - cell * LPC;
      const int ty = cell / TW, tx = cell - ty * TW;
      const int y = min(max(y0 + ty - KH / 2, 0), H - 1), x = min(max(x0 + tx - KW / 2, 0), W - 1);
      r[j] = *(const V*)(F + ((size_t)y * W + x) * PBD_FLEN + q * EPV);
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = tid + (j0 + j) * 256;
      if (i < N) {
        const int cell = i / LPC, q = i - cell * LPC;
        const int ty = cell / TW, tx = cell - ty * TW;
        const int y = y0 + ty - KH / 2, x = x0 + tx - KW / 2;
        const bool inside = (y >= 0 && y < H && x >= 0 && x < W);
        T* d = ft + cell * CSTR + q * EPV;
#pragma unroll
        for (int k = 0; k < EPV; ++k) d[k] = inside ? r[j].e[k] : (T)((q == LPC - 1 && k == EPV - 1) ? 1 : 0);
      }
    }
  }
}

// T = float: SpatialConvolutionEngine(CV_32F); T = double: CV_64F with the filters converted to double
// (src/PartsBasedDetector.cpp:110-117) — wT holds them as T.
template <typename T, int KH, int KW>
__global__ __launch_bounds__(256) void k_conv_exact(const ConvTile* __restrict__ tiles,
                                                    const LevelDev* __restrict__ levels,
                                                    const T* __restrict__ feat, const T* __restrict__ wT,
                                                    T* __restrict__ resp, int nf, int nfpad, int groups_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* ft = (T*)smem;  // [(CT+KH-1)][(CT+KW-1)][CSTR]
  const ConvTile t = tiles[blockIdx.x];
  const LevelDev lv = levels[t.level];
  const int H = lv.ch, W = lv.cw;
  const int TW = CT + KW - 1;
  const int tid = threadIdx.x;
  const T* F = feat + lv.cell_off * PBD_FLEN;
  // stage the tile: 8 lanes x float4 (16 x double2) per cell -> coalesced 128 B (256 B) per cell
  if constexpr (sizeof(T) == 4) stage_feature_tile<KH, KW>(ft, F, t.y0, t.x0, H, W, tid);
  else stage_feature_tile_t<T, KH, KW>(ft, F, t.y0, t.x0, H, W, tid);
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int oy = t.y0 + ly, ox = t.x0 + lx;
  const bool valid = (oy < H && ox < W);
  const T* fbase = ft + (ly * TW + lx) * CSTR;
  T* R = resp + lv.cell_off * nf;  // level base, plane n at + n*H*W
  const int g0 = blockIdx.y * groups_per_wg;
  for (int g = g0; g < g0 + groups_per_wg; ++g) {
    const int n0 = g * NFG;
    if (n0 >= nf) break;
    T tot[NFG];
#pragma unroll
    for (int n = 0; n < NFG; ++n) tot[n] = (T)0;
    for (int c = 0; c < PBD_FLEN; ++c) {
      T acc[NFG];
#pragma unroll
      for (int n = 0; n < NFG; ++n) acc[n] = (T)0;
#pragma unroll
      for (int i = 0; i < KH; ++i) {
#pragma unroll
        for (int j = 0; j < KW; ++j) {
          const T f = fbase[(i * TW + j) * CSTR + c];
          const T* w = wT + ((size_t)(i * KW + j) * PBD_FLEN + c) * nfpad + n0;  // wave-uniform
#pragma unroll
          for (int n = 0; n < NFG; ++n) acc[n] += w[n] * f;
        }
      }
#pragma unroll
      for (int n = 0; n < NFG; ++n) tot[n] += acc[n];
    }
    if (valid) {
#pragma unroll
      for (int n = 0; n < NFG; ++n)
        if (n0 + n < nf) R[(size_t)(n0 + n) * H * W + (size_t)oy * W + ox] = tot[n];
    }
  }
}

// double instantiation of the exact filter bank.  A 20x20-cell tile of 32 doubles is 105 KB of LDS (one
// workgroup per CU), so the tile is staged in two 16-channel halves (53 KB: three workgroups per CU).  The
// reference's order survives: channels are still visited 0..31, each channel's tap chain starts from
// zero and the channel partials are added to the running total in channel order (`pdf += pdfc`, :92);
// the totals of the workgroup's GPW filter groups simply stay in registers across the two halves.
#define CHALF 16     // channels staged per pass
#define CSTRH 17     // LDS doubles per cell and pass (16 + 1 pad)
template <int KH, int KW, int GPW>
__global__ __launch_bounds__(256) void k_conv_exact_f64(const ConvTile* __restrict__ tiles,
                                                        const LevelDev* __restrict__ levels,
                                                        const double* __restrict__ feat, const double* __restrict__ wT,
                                                        double* __restrict__ resp, int nf, int nfpad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* ft = (double*)smem;  // [(CT+KH-1)][(CT+KW-1)][CSTRH]
  const ConvTile t = tiles[blockIdx.x];
  const LevelDev lv = levels[t.level];
  const int H = lv.ch, W = lv.cw;
  constexpr int TW = CT + KW - 1, TH = CT + KH - 1;
  const int tid = threadIdx.x;
  const double* F = feat + lv.cell_off * PBD_FLEN;
  const int ly = tid >> 4, lx = tid & 15;
  const int oy = t.y0 + ly, ox = t.x0 + lx;
  const bool valid = (oy < H && ox < W);
  const double* fbase = ft + (ly * TW + lx) * CSTRH;
  double* R = resp + lv.cell_off * nf;
  const int g0 = blockIdx.y * GPW;
  double tot[GPW][NFG];
#pragma unroll
  for (int g = 0; g < GPW; ++g)
#pragma unroll
    for (int n = 0; n < NFG; ++n) tot[g][n] = 0.0;
  for (int half = 0; half < PBD_FLEN / CHALF; ++half) {
    if (half) __syncthreads();   // everyone is done with the previous half
    {  // stage 16 channels of every cell: 8 lanes x double2 per cell, batches of independent loads
      constexpr int LPC = CHALF / 2, N = TH * TW * LPC, NB = (N + 255) / 256, BATCH = 7;
      for (int j0 = 0; j0 < NB; j0 += BATCH) {
        double2 r[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int i = min(tid + (j0 + j) * 256, N - 1);
          const int cell = i / LPC, q = i - cell * LPC;
          const int ty = cell / TW, tx = cell - ty * TW;
          const int y = min(max(t.y0 + ty - KH / 2, 0), H - 1), x = min(max(t.x0 + tx - KW / 2, 0), W - 1);
          r[j] = *(const double2*)(F + ((size_t)y * W + x) * PBD_FLEN + half * CHALF + q * 2);
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int i = tid + (j0 + j) * 256;
          if (i < N) {
            const int cell = i / LPC, q = i - cell * LPC;
            const int ty = cell / TW, tx = cell - ty * TW;
            const int y = t.y0 + ty - KH / 2, x = t.x0 + tx - KW / 2;
            double2 v = r[j];
            if (!(y >= 0 && y < H && x >= 0 && x < W))     // border 0, 1 for the truncation channel (:147-155)
              v = make_double2(0.0, (half == PBD_FLEN / CHALF - 1 && q == LPC - 1) ? 1.0 : 0.0);
            double* d = ft + cell * CSTRH + q * 2;
            d[0] = v.x; d[1] = v.y;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
      const int n0 = (g0 + g) * NFG;
      if (n0 >= nf) break;
      for (int cc = 0; cc < CHALF; ++cc) {
        const int c = half * CHALF + cc;
        double acc[NFG];
#pragma unroll
        for (int n = 0; n < NFG; ++n) acc[n] = 0.0;
#pragma unroll
        for (int i = 0; i < KH; ++i) {
#pragma unroll
          for (int j = 0; j < KW; ++j) {
            const double f = fbase[(i * TW + j) * CSTRH + cc];
            const double* w = wT + ((size_t)(i * KW + j) * PBD_FLEN + c) * nfpad + n0;  // wave-uniform
#pragma unroll
            for (int n = 0; n < NFG; ++n) acc[n] += w[n] * f;
          }
        }
#pragma unroll
        for (int n = 0; n < NFG; ++n) tot[g][n] += acc[n];
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
      const int n0 = (g0 + g) * NFG;
#pragma unroll
      for (int n = 0; n < NFG; ++n)
        if (n0 + n < nf) R[(size_t)(n0 + n) * H * W + (size_t)oy * W + ox] = tot[g][n];
    }
  }
}

// generic-size fallback (runtime kh, kw <= 9)
template <typename T>
__global__ __launch_bounds__(256) void k_conv_exact_generic(const ConvTile* __restrict__ tiles,
                                                            const LevelDev* __restrict__ levels,
                                                            const T* __restrict__ feat,
                                                            const T* __restrict__ wT, T* __restrict__ resp,
                                                            int nf, int nfpad, int KH, int KW) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* ft = (T*)smem;
  const ConvTile t = tiles[blockIdx.x];
  const LevelDev lv = levels[t.level];
  const int H = lv.ch, W = lv.cw;
  const int TW = CT + KW - 1, TH = CT + KH - 1;
  const int tid = threadIdx.x;
  const T* F = feat + lv.cell_off * PBD_FLEN;
  for (int i = tid; i < TH * TW * PBD_FLEN; i += 256) {
    const int cell = i >> 5, c = i & 31;
    const int ty = cell / TW, tx = cell - ty * TW;
    const int y = t.y0 + ty - KH / 2, x = t.x0 + tx - KW / 2;
    const bool inside = (y >= 0 && y < H && x >= 0 && x < W);
    const T v = F[((size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * PBD_FLEN + c];
    ft[cell * CSTR + c] = inside ? v : (T)(c == PBD_FLEN - 1 ? 1 : 0);
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int oy = t.y0 + ly, ox = t.x0 + lx;
  const bool valid = (oy < H && ox < W);
  const T* fbase = ft + (ly * TW + lx) * CSTR;
  T* R = resp + lv.cell_off * nf;
  for (int n = blockIdx.y; n < nf; n += gridDim.y) {
    T tot = (T)0;
    for (int c = 0; c < PBD_FLEN; ++c) {
      T acc = (T)0;
      for (int i = 0; i < KH; ++i)
        for (int j = 0; j < KW; ++j)
          acc += wT[((size_t)(i * KW + j) * PBD_FLEN + c) * nfpad + n] * fbase[(i * TW + j) * CSTR + c];
      tot += acc;
    }
    if (valid) R[(size_t)n * H * W + (size_t)oy * W + ox] = tot;
  }
}

template <typename T>
static void launch_conv_exact_t(const ConvTile* tiles, int ntiles, const LevelDev* levels, const T* feat,
                                const T* wT, T* resp, int nf, int nfpad, int kh, int kw, hipStream_t s) {
  const size_t lds = sizeof(T) * (CT + kh - 1) * (CT + kw - 1) * CSTR;
  if constexpr (sizeof(T) == 8) {
    if (kh == 5 && kw == 5) {
      constexpr int GPW = 4;
      const size_t ldsh = sizeof(double) * (CT + 4) * (CT + 4) * CSTRH;
      static LdsOptIn optin64;
      optin64.ensure((const void*)k_conv_exact_f64<5, 5, GPW>, ldsh);
      const int groups = (nf + NFG - 1) / NFG;
      dim3 grid(ntiles, (groups + GPW - 1) / GPW);
      hipLaunchKernelGGL((k_conv_exact_f64<5, 5, GPW>), grid, dim3(256), ldsh, s, tiles, levels, feat, wT, resp, nf, nfpad);
      return;
    }
  }
  if (kh == 5 && kw == 5) {
    static LdsOptIn optin;   // one per instantiation
    optin.ensure((const void*)k_conv_exact<T, 5, 5>, lds);
    const int groups = (nf + NFG - 1) / NFG;
    const int gpw = 4;
    dim3 grid(ntiles, (groups + gpw - 1) / gpw);
    hipLaunchKernelGGL((k_conv_exact<T, 5, 5>), grid, dim3(256), lds, s, tiles, levels, feat, wT, resp, nf, nfpad, gpw);
  } else {
    static LdsOptIn opting;
    opting.ensure((const void*)k_conv_exact_generic<T>, lds);
    dim3 grid(ntiles, nf < 16 ? nf : 16);
    hipLaunchKernelGGL(k_conv_exact_generic<T>, grid, dim3(256), lds, s, tiles, levels, feat, wT, resp, nf, nfpad, kh, kw);
  }
}

// ts = sizeof(T): 4 -> SpatialConvolutionEngine(CV_32F), 8 -> CV_64F
void launch_conv_exact(const ConvTile* tiles, int ntiles, const LevelDev* levels, const void* feat, const void* wT,
                       void* resp, int ts, int nf, int nfpad, int kh, int kw, hipStream_t s) {
  if (ntiles <= 0) return;
  if (ts == 8) launch_conv_exact_t<double>(tiles, ntiles, levels, (const double*)feat, (const double*)wT, (double*)resp, nf, nfpad, kh, kw, s);
  else launch_conv_exact_t<float>(tiles, ntiles, levels, (const float*)feat, (const float*)wT, (float*)resp, nf, nfpad, kh, kw, s);
}

// ---------------------------------------------------------------------------
// fp32 MFMA implicit GEMM (v_mfma_f32_32x32x2_f32), M = cells, N = filters, K = kh*kw*32.
// Workgroup = 256 threads = 4 waves: tile = 16x16 cells (M = 256) x ONE 32-filter n-tile; the
// grid is (tiles, nfpad/32), so work units are small (3 resident per CU, ~12 per CU for the person
// model) and the tail of the launch is short.  Wave w owns cell rows 4w..4w+3 = two 32-row MFMA
// M-tiles (2 cell rows x 16 cols each): 2 accumulators of 16 VGPRs.
//  * A (features): the 20x20-cell tile with halo is staged once in LDS, cell stride 33 floats, so
//    the 32 lanes of an M-tile read conflict-free; lane l holds A[i = l&31][k = l>>5].
//  * B (weights, [tap][channel][nfpad]): 512 KB for the whole bank, L2-resident.  A lane's B
//    operand is ONE float per MFMA (B[k = l>>5][j = l&31]); the 16 values of a tap are loaded
//    straight from L2 into registers a whole tap (2048 MFMA cycles) ahead of use — no weight LDS,
//    no barrier anywhere in the K loop.
//  * K order: tap-major, channel-minor; the accumulation is a k-ordered fp32 fma chain.
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KH, int KW>
__global__ __launch_bounds__(256) void k_conv_mfma(const ConvTile* __restrict__ tiles,
                                                   const LevelDev* __restrict__ levels,
                                                   const float* __restrict__ feat, const float* __restrict__ wT,
                                                   float* __restrict__ resp, int nf, int nfpad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TW = CT + KW - 1;
  float* ft = (float*)smem;                 // [TH][TW][CSTR]
  CONV_STAMP(0);
  const ConvTile t = tiles[blockIdx.x];
  const LevelDev lv = levels[t.level];
  const int H = lv.ch, W = lv.cw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbase = blockIdx.y * 32;
  const float* F = feat + lv.cell_off * PBD_FLEN;
  // A operand: lane l holds A[i = l&31][k = l>>5]; M-tile m of this wave: cell rows 4*wave + 2*m + (ai>>4), col ai&15
  const int ai = lane & 31, ak = lane >> 5;
  // B operand: B[k = l>>5][j = l&31] -> wT[(tap*32 + c + ak)*nfpad + nbase + (l&31)]
  const float* bsrc = wT + (size_t)ak * nfpad + nbase + (lane & 31);
  // two register sets in explicit ping-pong (the tap loop is unrolled by two): while one set feeds
  // the 32 MFMAs of a tap, the other receives the next tap's 16 values.  With a single pair of
  // arrays and a copy hipcc merges them and ends up loading the next tap AFTER the last MFMA that
  // reads the registers, then waits vmcnt(0) at the loop tail: a full L2 round trip per tap.
  float b0[16], b1[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) b0[u] = bsrc[(size_t)(2 * u) * nfpad];  // tap 0, issued before the tile staging
  stage_feature_tile<KH, KW>(ft, F, t.y0, t.x0, H, W, tid);
  __syncthreads();
  CONV_STAMP(1);

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const int arow0 = 4 * wave + (ai >> 4), acol = ai & 15;
  const float* abase0 = ft + (arow0 * TW + acol) * CSTR + ak;
  const float* abase1 = ft + ((arow0 + 2) * TW + acol) * CSTR + ak;
  constexpr int NTAP = KH * KW;

  auto load_tap = [&](float (&dst)[16], int tap) {
    const float* bs = bsrc + (size_t)min(tap, NTAP - 1) * PBD_FLEN * nfpad;
#pragma unroll
    for (int u = 0; u < 16; ++u) dst[u] = bs[(size_t)(2 * u) * nfpad];
  };
  auto mma_tap = [&](const float (&bw)[16], int tap) {
    const int ti = tap / KW, tj = tap - ti * KW;
    const float* a0 = abase0 + (ti * TW + tj) * CSTR;
    const float* a1 = abase1 + (ti * TW + tj) * CSTR;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float av0 = a0[2 * u], av1 = a1[2 * u];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bw[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bw[u], acc1, 0, 0, 0);
    }
  };
  for (int tap = 0; tap < NTAP; tap += 2) {
    load_tap(b1, tap + 1);
    mma_tap(b0, tap);
    if (tap + 1 < NTAP) {
      load_tap(b0, tap + 2);
      mma_tap(b1, tap + 1);
    }
  }
  CONV_STAMP(2);
  __syncthreads();  // all waves are done reading the feature tile: reuse it for the epilogue
  CONV_STAMP(3);
  // Epilogue.  C/D layout 32x32: col(j) = lane&31, row(i) = (reg&3) + 8*(reg>>2) + 4*(lane>>5),
  // i.e. a lane holds ONE filter and 16 scattered cells: storing that directly would be 4-byte
  // scatters across 32 response planes.  Transpose the wave's 64-cell x 32-filter slab through the
  // (now free) feature-tile LDS so lanes run along cells: every store instruction then writes
  // four 64-B row segments of one plane.
  float* R = resp + lv.cell_off * nf;
  float* tr = ft + wave * (32 * 65);  // per-wave [32 filters][64 cells + 1]
  const int py = t.y0 + 4 * wave + (lane >> 4), pxx = t.x0 + (lane & 15);
  const bool pvalid = (py < H && pxx < W);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    tr[(lane & 31) * 65 + i] = acc0[r];
    tr[(lane & 31) * 65 + 32 + i] = acc1[r];
  }
  __syncthreads();
  for (int j = 0; j < 32; ++j) {
    const int fn = nbase + j;
    if (fn < nf && pvalid) R[(size_t)fn * H * W + (size_t)py * W + pxx] = tr[j * 65 + lane];
  }
  CONV_STAMP(4);
}

void launch_conv_mfma(const ConvTile* tiles, int ntiles, const LevelDev* levels, const float* feat,
                      const float* wT, float* resp, int nf, int nfpad, int kh, int kw, hipStream_t s) {
  if (ntiles <= 0) return;
  if (kh != 5 || kw != 5) { launch_conv_exact(tiles, ntiles, levels, feat, wT, resp, 4, nf, nfpad, kh, kw, s); return; }
  const size_t lds = sizeof(float) * (CT + 4) * (CT + 4) * CSTR;
  static LdsOptIn optin;
  optin.ensure((const void*)k_conv_mfma<5, 5>, lds);
  dim3 grid(ntiles, (nf + 31) / 32);
  hipLaunchKernelGGL((k_conv_mfma<5, 5>), grid, dim3(256), lds, s, tiles, levels, feat, wT, resp, nf, nfpad);
}

// ---------------------------------------------------------------------------
// 16x16x4 MFMA implicit GEMM, instantiated for double (v_mfma_f64_16x16x4_f64: the filter bank of the
// double instantiation) and for float (v_mfma_f32_16x16x4_f32).  M = cells, N = filters, K = kh*kw*32.
// Measured on MI355X (tests/tools/mfma64_probe.hip, mfma16_probe.hip): f64 64 cycles per instruction and
// SIMD = 72 TFLOP/s; operand layout A[i = l&15][k = l>>4], B[k = l>>4][j = l&15] for both; result
// D[i = 4*reg + (l>>4)][j = l&15] (f64) / D[i = 4*(l>>4) + reg][j = l&15] (f32).
// Workgroup = 4 waves: 16x16 cells x ONE 16-filter n-tile, grid = (tiles, nfpad/16).  The VALID cells of the tile
// (levels are ragged: the last tile of a row / column is cut by the level's edge) are numbered row-major and cut
// into 16-cell M-tiles; wave w owns M-tiles w, w + 4, w + 8, w + 12 = up to four accumulators, and an M-tile beyond
// the last valid cell issues no MFMAs (13 % of the MFMA work of a 640x480 pyramid was padding when an M-tile was a
// fixed 16-cell row segment).  The 20x20-cell feature tile is staged in
// NHALF channel groups (double: two 16-channel halves, 54 KB -> three workgroups per CU), cell stride
// CH+1 elements (conflict-free across the 16 cells of an M-tile).  B: one element per lane and k-step, a
// whole tap loaded from the L2-resident [tap][channel][nfpad] array one tap ahead, ping-pong registers.
// Accumulation is a k-ordered fma chain (half, tap, channel): not the reference's order, tolerance-based.
// ---------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Mfma16;
template <> struct Mfma16<double> {
  typedef f64x4 acc_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int drow(int reg, int ak) { return 4 * reg + ak; }
};
template <> struct Mfma16<float> {
  typedef f32x4 acc_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int drow(int reg, int ak) { return 4 * ak + reg; }
};

// B4 (float, two channel halves; double, four 8-channel groups): wT is the [tap][group][k][nfpad][u] copy of the filters (channel
// CH group + 4 u + k): a lane reads the k-steps of a tap and n-tile with ONE 16-byte load instead of four global_load_dword (which cost the MFMA pipe a quarter of its
// issue rate with two waves per SIMD: tests/tools/mfma_rate_probe.hip)
// KH_T / KW_T > 0: compile-time filter size (the 5x5 bank of the person / face models: tap loops and tile geometry fold);
// 0: the size comes from the kernel arguments (any kh x kw <= 9 x 9, src/SpatialConvolutionEngine.cpp:133-159 takes any).
template <typename T, int KH_T, int KW_T, int NHALF, int WPE, int NTW = 1, bool B4 = false>   // WPE: waves per SIMD the register allocation must allow; NTW: 16-filter n-tiles per workgroup
__global__ __launch_bounds__(256, WPE) void k_conv_mfma16(const ConvTile* __restrict__ tiles,
                                                     const LevelDev* __restrict__ levels,
                                                     const T* __restrict__ feat, const T* __restrict__ wT,
                                                     T* __restrict__ resp, int nf, int nfpad, int ntiles_total, int kh_rt, int kw_rt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Mfma16<T> MM;
  const int KH = KH_T > 0 ? KH_T : kh_rt, KW = KW_T > 0 ? KW_T : kw_rt;
  const int TW = CT + KW - 1, TH = CT + KH - 1, NTAP = KH * KW;
  // channels per pass, LDS cell stride, k-steps per tap.  Float: stride CH + 2 = 18 dwords: the 32 lanes of one LDS
  // access group (16 cells x 2 channels) then hit 32 different banks (16 * 18 mod 32 are the 16 even residues); with
  // 17 the cell 15 / channel 1 lane fell on cell 0's bank (SQ_LDS_BANK_CONFLICT was twice SQ_ACTIVE_INST_LDS)
  constexpr int CH = PBD_FLEN / NHALF, CS = CH + (sizeof(T) == 4 ? 2 : 1), KS = CH / 4;
  constexpr int EPV = 16 / (int)sizeof(T), LPC = CH / EPV;         // elements per 16-byte vector, lanes per cell
  struct alignas(16) V { T e[EPV]; };
  T* ft = (T*)smem;                         // [TH][TW][CS]
  CONV_STAMP(0);
#ifdef PBD_PROBES
  {  // probe: issue priority by workgroup index, to pull co-resident workgroups out of phase
    const unsigned lin_ = blockIdx.x + blockIdx.y * gridDim.x;
    const int mode = nfpad >> 16;
    const unsigned pr = mode == 1 ? (lin_ & 3u) : mode == 2 ? ((lin_ >> 8) & 3u) : mode == 3 ? ((lin_ >> 3) & 3u) : mode == 4 ? ((lin_ >> 10) & 3u) : mode == 5 ? (blockIdx.y & 3u) : 0u;
    if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
  nfpad &= 0xffff;
#endif
  // XCD-aware workgroup -> (tile, n-tile) mapping.  Workgroup b runs on XCD b % 8 and every XCD has its own L2; the
  // ny n-tile workgroups of one spatial tile all stage the same 20x20-cell feature tile.  With (tile, n-tile) =
  // (blockIdx.x, blockIdx.y) they were 604 workgroups apart and on 8 different XCDs: the tile came from HBM ~10
  // times (FETCH 5.9x the algorithmic bytes, r01).  Here groups of 8 tiles x ny n-tiles are laid out so that all
  // n-tiles of a tile share b % 8 and are dispatched within 8 * ny consecutive workgroups: one HBM fetch, ny - 1 L2 hits.
  const int ny = gridDim.y;                      // n-tiles (the launch keeps the 2-D grid shape; only the roles are permuted)
  const int lin = blockIdx.x + blockIdx.y * gridDim.x;
  const int grp = lin / (8 * ny), rem = lin - grp * (8 * ny);
  const int tile_i = grp * 8 + (rem & 7), ntile_i = rem >> 3;
  if (tile_i >= ntiles_total) return;            // the grid is padded to a multiple of 8 tiles
  const ConvTile t = tiles[tile_i];
  const LevelDev lv = levels[t.level];
  const int H = lv.ch, W = lv.cw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbase = ntile_i * (16 * NTW);
  const T* F = feat + lv.cell_off * PBD_FLEN;
  const int ai = lane & 15, ak = lane >> 4;
  static_assert(!B4 || (sizeof(T) == 4 && NHALF == 2) || (sizeof(T) == 8 && NHALF == 4), "16-byte B loads: the k-steps of a channel group fill one 16-byte vector");
  const T* bsrc = B4 ? wT + ((size_t)ak * nfpad + nbase + ai) * KS     // w4[tap 0][group 0][k = ak][filter nbase + ai][u = 0..KS-1]
                     : wT + (size_t)ak * nfpad + nbase + ai;           // B[k = ak][j = ai] of k-step 0, tap 0, half 0, n-tile 0 (n-tile nt: + 16 nt)
  typename MM::acc_t acc[NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[nt][m][r] = (T)0;
  // packed M-tiles: valid cell c = 16 * (wave + 4 m) + ai of the vh x vw valid region -> (c / vw, c % vw); cells past the
  // last one repeat it (their products are never stored)
  const int vw = min(CT, W - t.x0), vh = min(CT, H - t.y0), ncell = vw * vh;
  const int nmt = (ncell + 15) >> 4;                                                        // M-tiles of the tile
  const int mvalid = __builtin_amdgcn_readfirstlane(max(0, min(4, (nmt - wave + 3) >> 2)));   // M-tiles of this wave
  // c / vw for c < 256, vw <= 16 as a multiply: floor(c * ceil(2^16 / vw) / 2^16) is exact there (error < c (vw - 1) / (vw 2^16) < 1 / vw);
  // one wave-uniform division for the constant instead of a full 32-bit division sequence per lane and M-tile
  const unsigned vw_magic = 65535u / (unsigned)vw + 1u;
  int aoff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = min(16 * (wave + 4 * m) + ai, ncell - 1);
    const int cy = (int)(((unsigned)c * vw_magic) >> 16), cx = c - cy * vw;
    aoff[m] = (cy * TW + cx) * CS + ak;
  }

#pragma unroll 1
  for (int half = 0; half < NHALF; ++half) {
    if (half) __syncthreads();
    const T* bh = bsrc + (size_t)(half * CH) * nfpad;
    T b0[NTW][KS], b1[NTW][KS];
    auto load_tap = [&](T (&dst)[NTW][KS], int tap) {
      const T* bs = bh + (size_t)min(tap, NTAP - 1) * PBD_FLEN * nfpad;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        if constexpr (B4) {
          const V w = *(const V*)(bs + 16 * KS * nt);
#pragma unroll
          for (int u = 0; u < KS; ++u) dst[nt][u] = w.e[u];
        } else {
#pragma unroll
          for (int u = 0; u < KS; ++u) dst[nt][u] = bs[(size_t)(4 * u) * nfpad + 16 * nt];
        }
      }
    };
    load_tap(b0, 0);   // tap 0, issued before the staging
    {  // stage CH channels of every cell: LPC lanes x 16 B per cell, batches of independent loads
      const int N = TH * TW * LPC, NB = (N + 255) / 256;
      constexpr int BATCH = WPE >= 4 ? 4 : 7;   // (a tighter register allocation stages in smaller batches)
      for (int j0 = 0; j0 < NB; j0 += BATCH) {
        V r[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int i = min(tid + (j0 + j) * 256, N - 1);
          const int cell = i / LPC, q = i - cell * LPC;
          const int ty = cell / TW, tx = cell - ty * TW;
          const int y = min(max(t.y0 + ty - KH / 2, 0), H - 1), x = min(max(t.x0 + tx - KW / 2, 0), W - 1);
          r[j] = *(const V*)(F + ((size_t)y * W + x) * PBD_FLEN + half * CH + q * EPV);
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int i = tid + (j0 + j) * 256;
          if (i < N) {
            const int cell = i / LPC, q = i - cell * LPC;
            const int ty = cell / TW, tx = cell - ty * TW;
            const int y = t.y0 + ty - KH / 2, x = t.x0 + tx - KW / 2;
            const bool inside = (y >= 0 && y < H && x >= 0 && x < W);
            T* d = ft + cell * CS + q * EPV;
#pragma unroll
            for (int k = 0; k < EPV; ++k)   // border 0, 1 for the truncation channel (:147-155)
              d[k] = inside ? r[j].e[k] : (T)((half == NHALF - 1 && q == LPC - 1 && k == EPV - 1) ? 1 : 0);
          }
        }
      }
    }
    __syncthreads();
    CONV_STAMP(1 + 2 * half);
    // The K loop of one channel group, instantiated per number of M-tiles the wave owns (MV = 1..4, wave-uniform: a ragged
    // tile leaves some waves with fewer).  With the count tested inside the loop (`if (m < mvalid)`) hipcc guarded EVERY MFMA
    // with its own scalar branch — 32 branches per tap between instructions that should issue back to back.
    auto k_loop = [&](auto mv_tag) {
      constexpr int MV = decltype(mv_tag)::value;
      auto mma_tap = [&](const T (&bw)[NTW][KS], int tap) {
        const int ti = tap / KW, tj = tap - ti * KW;
        const T* a = ft + (ti * TW + tj) * CS;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
          T av[MV];
#pragma unroll
          for (int m = 0; m < MV; ++m) av[m] = a[aoff[m] + 4 * u];    // one A element per M-tile, shared by the workgroup's n-tiles
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int m = 0; m < MV; ++m) acc[nt][m] = MM::mma(av[m], bw[nt][u], acc[nt][m]);
        }
      };
      auto tap_pair = [&](int tap) {
        load_tap(b1, tap + 1);
        mma_tap(b0, tap);
        if (tap + 1 < NTAP) {
          load_tap(b0, tap + 2);
          mma_tap(b1, tap + 1);
        }
      };
      if constexpr (NHALF == 1 && NTW == 1 && KH_T > 0) {
        for (int tap = 0; tap < NTAP; tap += 2) tap_pair(tap);
      } else {   // with half the k-steps per tap hipcc would unroll all taps and run out of registers
        _Pragma("unroll 1") for (int tap = 0; tap < NTAP; tap += 2) tap_pair(tap);
      }
    };
    if (mvalid == 4) k_loop(std::integral_constant<int, 4>());
    else if (mvalid == 3) k_loop(std::integral_constant<int, 3>());
    else if (mvalid == 2) k_loop(std::integral_constant<int, 2>());
    else if (mvalid == 1) k_loop(std::integral_constant<int, 1>());   // (0: M-tiles past the tile's last valid cell: no MFMA work)
    CONV_STAMP(2 + 2 * half);
  }
  __syncthreads();  // all waves are done reading the feature tile: reuse it for the epilogue
  CONV_STAMP(5);
  // Epilogue: transpose the wave's 64-cell x 16-filter slab of each n-tile through LDS so lanes run along cells
  // (a store instruction then writes whole 64-B row segments of one response plane).  The slab is private to the
  // wave and a wave's LDS operations execute in order, so the n-tiles simply follow each other.
  T* R = resp + lv.cell_off * nf;
  T* tr = ft + wave * (16 * 65);           // per-wave [16 filters][64 cells + 1]
  // lane -> slot (M-tile lane >> 4, row lane & 15) -> packed cell -> (row, column) of the level
  const int pc = 16 * (wave + 4 * (lane >> 4)) + (lane & 15);
  const int pcy = (int)(((unsigned)pc * vw_magic) >> 16), py = t.y0 + pcy, pxx = t.x0 + (pc - pcy * vw);   // (pc < 256)
  const bool pvalid = pc < ncell;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) tr[ai * 65 + m * 16 + MM::drow(r, ak)] = acc[nt][m][r];   // D[i][j = ai] of M-tile m
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // (plane base wave-uniform, the lane's cell a 32-bit offset: a store is one LDS read + one store instruction, no address
    // arithmetic per filter — it was seven vector instructions per store)
    if (pvalid) {
      const unsigned cellb = (unsigned)(py * W + pxx) * (unsigned)sizeof(T);      // < 2^31 (plan_frame: a level has < 2^28 cells)
      for (int j = 0; j < 16; ++j) {
        const int fn = nbase + 16 * nt + j;
        if (fn < nf) {
          char* plane = (char*)(R + (size_t)fn * H * W);
          *(T*)(plane + cellb) = tr[j * 65 + lane];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  CONV_STAMP(6);
}

int g_conv_lds_req_kb = 0;   // set by pbd_api.cpp from PBD_CONV_LDS_KB in probe / tuning builds
// KH_T = KW_T = 5: the compile-time 5x5 instantiation; 0: any kh x kw (run-time tap loop)
template <typename T, int NHALF, int WPE, int NTW = 1, bool B4 = false, int KH_T = 5, int KW_T = 5>
static void launch_conv_mfma16_t(const ConvTile* tiles, int ntiles, const LevelDev* levels, const T* feat,
                                 const T* wT, T* resp, int nf, int nfpad, hipStream_t s, int kh = 5, int kw = 5) {
  size_t lds = std::max(sizeof(T) * (CT + kh - 1) * (CT + kw - 1) * (PBD_FLEN / NHALF + (sizeof(T) == 4 ? 2 : 1)), sizeof(T) * 4 * 16 * 65);
  if (g_conv_lds_req_kb > 0) lds = std::max(lds, (size_t)g_conv_lds_req_kb * 1024);   // tuning builds: occupancy cap by LDS request
  static LdsOptIn optin;   // one per instantiation
  optin.ensure((const void*)k_conv_mfma16<T, KH_T, KW_T, NHALF, WPE, NTW, B4>, lds);
  dim3 grid((ntiles + 7) / 8 * 8, (nf + 16 * NTW - 1) / (16 * NTW));   // tiles padded to a multiple of 8 (XCD-aware mapping in the kernel)
  static const int prio_mode = PBD_PROBE_ENV("PBD_CONV_PRIO") ? atoi(PBD_PROBE_ENV("PBD_CONV_PRIO")) : 0;   // probe build only
  hipLaunchKernelGGL((k_conv_mfma16<T, KH_T, KW_T, NHALF, WPE, NTW, B4>), grid, dim3(256), lds, s, tiles, levels, feat, wT, resp, nf, nfpad | (prio_mode << 16), ntiles, kh, kw);
}

// ---------------------------------------------------------------------------
// k_conv_glds: the fp32 filter bank as a PERSISTENT, double-buffered workgroup.  k_conv_mfma16 runs the MFMA pipe at
// ~93 % while its K loops run, but every workgroup first stages its tile (global -> registers -> LDS, 9 + 5 us of a
// 75 us life) and ends with an epilogue, and co-resident workgroups run those phases in step: over the whole kernel
// the pipe is ~66 % busy.  Here a workgroup loops over work units (tile, pair of 16-filter n-tiles) and the NEXT
// channel half (of this unit, or half 0 of the next unit) streams into the other LDS buffer with
// global_load_lds_dwordx4 (LDS-DMA: no staging registers, no ds_write pass) while the MFMAs of the current half
// run.  LDS image of a half: [cell 0..399][16 channels], 64 B per cell, lane-linear as the DMA writes it (piece p =
// cells 16 p .. 16 p + 15, lane = 4 (cell & 15) + 16-byte chunk); border cells are DMA'd from a constant cell
// (0, and 1 for the truncation channel 31, src/SpatialConvolutionEngine.cpp:147-155).  A operand: lane (i, k) reads ONE
// ds_read_b128 per M-tile and tap = channels 4k .. 4k+3 of its cell, which feed k-steps s = 0..3 (k-step s contracts
// channels {s, 4+s, 8+s, 12+s}; the B rows are picked to match) -- the 16 cells of a full-width M-tile are 1 KB
// contiguous: conflict-free without padding.  Units of XCD x: tile positions 8 g + x (same convention as
// k_conv_mfma16, so the plan's neighbour pairing holds), n-pairs minor: the n-pairs of one tile are taken by adjacent
// workgroups of the XCD at the same time (one HBM fetch of the tile, L2 hits for the others).
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void pbd_lds_void;
typedef __attribute__((address_space(1))) const void pbd_glb_cvoid;

// PERSIST = false: the same operand paths (LDS-DMA staging, 16-byte A and B reads) without the persistent loop: one unit per
// workgroup, ONE 25.6 KB buffer (stage half, barrier, K loop, barrier, ...), grid and XCD mapping of k_conv_mfma16.
template <int WPE, bool PERSIST = true>
__global__ __launch_bounds__(256, WPE) void k_conv_glds(const ConvTile* __restrict__ tiles, const LevelDev* __restrict__ levels,
                                                        const float* __restrict__ feat, const float* __restrict__ wT,
                                                        float* __restrict__ resp, int nf, int nfpad, int ntiles,
                                                        const float* __restrict__ border) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TW = CT + 4, NCELL = TW * TW, NTAP = 25, NTW = 2, CH = 16, CELLB = CH * 4, BUFB = NCELL * CELLB, NPIECE = NCELL / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ai = lane & 15, ak = lane >> 4;
  const int np = (nf + 16 * NTW - 1) / (16 * NTW);
  // persistent: workgroup j of XCD blockIdx.x % 8 takes units j, j + nwx, ... of that XCD's tile positions xcd, xcd + 8, ...;
  // else: groups of 8 tiles x np n-pairs, all n-pairs of a tile on one XCD (k_conv_mfma16's mapping), one unit per workgroup
  const int xcd = blockIdx.x & 7;
  const int grp_ = (int)blockIdx.x / (8 * np), rem_ = (int)blockIdx.x - grp_ * (8 * np);
  const int j = PERSIST ? (int)(blockIdx.x >> 3) : grp_ * np + (rem_ >> 3);
  const int nwx = PERSIST ? (int)(gridDim.x >> 3) : (1 << 30);
  const int ntx = ntiles > xcd ? (ntiles - xcd + 7) >> 3 : 0;     // tile positions xcd, xcd + 8, ...
  const int nunits = ntx * np;
  char* const buf0 = smem;
  char* const buf1 = smem + BUFB;

  // one wave's share of the LDS-DMA pieces of channel half `half` of the tile at (y0, x0) of a W x H level: 16 cells x 64 B per piece
  auto issue_stage = [&](int y0, int x0, int W, int H, size_t cell_off, int half, char* buf) {
    const float* F = feat + cell_off * PBD_FLEN + half * CH + 4 * (lane & 3);
    const float* bz = border + half * CH + 4 * (lane & 3);
    for (int p = wave; p < NPIECE; p += 4) {
      const int cell = 16 * p + (lane >> 2);
      const int ty = cell / TW, tx = cell - ty * TW;
      const int y = y0 + ty - 2, x = x0 + tx - 2;
      const bool inside = (y >= 0 && y < H && x >= 0 && x < W);
      const float* src = inside ? F + ((size_t)y * W + x) * PBD_FLEN : bz;
      __builtin_amdgcn_global_load_lds((pbd_glb_cvoid*)src, (pbd_lds_void*)(buf + p * 1024), 16, 0, 0);
    }
  };

  int v = j;
  if (v >= nunits) return;
  int y0, x0, W, H;
  size_t cell_off;
  {
    const ConvTile t = tiles[xcd + 8 * (v / np)];
    const LevelDev lv = levels[t.level];
    y0 = t.y0; x0 = t.x0; W = lv.cw; H = lv.ch; cell_off = lv.cell_off;
  }
  GLDS_INIT();
  GLDS_T(tl0);
  if (PERSIST) issue_stage(y0, x0, W, H, cell_off, 0, buf0);
  while (v < nunits) {
    GLDS_T(t0_);
    // the next unit's descriptor (after the last unit: this unit again, its half 0 is then re-staged into the free buffer —
    // the DMA issue stays unconditional: under a condition hipcc drains the whole load queue at every tap pair of the next K loop)
    const int vn = PERSIST ? v + nwx : nunits;
    int y0n, x0n, Wn, Hn;
    size_t cell_offn;
    {
      const ConvTile tn = tiles[xcd + 8 * ((vn < nunits ? vn : v) / np)];
      const LevelDev lvn = levels[tn.level];
      y0n = tn.y0; x0n = tn.x0; Wn = lvn.cw; Hn = lvn.ch; cell_offn = lvn.cell_off;
    }
    const int nbase = (v % np) * (16 * NTW);
    const int vw = min(CT, W - x0), vh = min(CT, H - y0), ncell = vw * vh;
    const int nmt = (ncell + 15) >> 4;
    const int mvalid = __builtin_amdgcn_readfirstlane(max(0, min(4, (nmt - wave + 3) >> 2)));
    int aoff[4];   // byte offset of the lane's 16-byte A chunk (tap 0) per M-tile
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int c = min(16 * (wave + 4 * m) + ai, ncell - 1);
      const int cy = c / vw, cx = c - cy * vw;
      aoff[m] = (cy * TW + cx) * CELLB + 16 * ak;
    }
    f32x4 acc[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nt][m][r] = 0.f;
    const float* bsrc = wT + ((size_t)ak * nfpad + nbase + ai) * 4;   // w4[tap 0][half 0][k = ak][filter nbase + ai][s = 0..3]

    // MV = 4: all four M-tiles of the wave hold valid cells (the common case: no branch in the K loop); MV = 0: decided per M-tile at run time
    auto kloop = [&](const char* buf, int half, auto mv_tag) {
      constexpr int MV = decltype(mv_tag)::value;
      const float* bh = bsrc + (size_t)half * 16 * nfpad;
      f32x4 b0[NTW], b1[NTW];
      f32x4 a0[4], a1[4];
      auto load_b = [&](f32x4 (&dst)[NTW], int tap) {
        const float* bs = bh + (size_t)tap * PBD_FLEN * nfpad;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) dst[nt] = *(const f32x4*)(bs + 64 * nt);
      };
      auto load_a = [&](f32x4 (&dst)[4], int tap) {
        const int ti = tap / 5, tj = tap - ti * 5;
        const char* a = buf + (ti * TW + tj) * CELLB;
#pragma unroll
        for (int m = 0; m < 4; ++m) dst[m] = *(const f32x4*)(a + aoff[m]);
      };
      auto mma = [&](const f32x4 (&av)[4], const f32x4 (&bw)[NTW]) {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (MV == 4 || m < mvalid) acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][s_], bw[nt][s_], acc[nt][m], 0, 0, 0);
      };
      load_b(b0, 0);
      load_a(a0, 0);
      // vmcnt(0): the wave's DMA pieces of the NEXT buffer (issued just before) and tap 0's B have landed.  hipcc cannot count
      // past an LDS-DMA in flight: left pending it waits vmcnt(0) at the first MFMA of every tap pair (exposing the B latency
      // 12 times per K loop); drained here once (~1 us, the co-resident workgroup's waves keep the pipe busy) the loop gets
      // exact counted waits.
      __builtin_amdgcn_s_waitcnt(0x0F70);
      // taps in pairs, operands in explicit ping-pong: the next tap's B (global, L2-resident) and A (LDS) are in flight while
      // this tap's 32 MFMAs issue.  No condition inside the loop (hipcc sinks loads into a conditional use); tap 24 is peeled.
      _Pragma("unroll 1") for (int tap = 0; tap < NTAP - 1; tap += 2) {
        load_b(b1, tap + 1);
        load_a(a1, tap + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_b(b0, tap + 2);
        load_a(a0, tap + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      mma(a0, b0);
    };

    // two channel halves through ONE K-loop instance: half h computes on buffer h while the DMA fills buffer 1 - h with
    // half 1 of this unit (h = 0) or half 0 of the next unit (h = 1)
    _Pragma("unroll 1") for (int half = 0; half < 2; ++half) {
      GLDS_T(ta_);
      if (!PERSIST) {
        if (half) __syncthreads();                        // every wave is done with half 0
        issue_stage(y0, x0, W, H, cell_off, half, buf0);
      }
      __syncthreads();          // buffer `half` has landed (the DMA queue is drained before the barrier); every wave is done with buffer 1 - half
      GLDS_T(tb_);
      GLDS_C(cb_);
      char* const cur = (PERSIST && half) ? buf1 : buf0;
      if (PERSIST) {
        if (half == 0) issue_stage(y0, x0, W, H, cell_off, 1, buf1);
        else issue_stage(y0n, x0n, Wn, Hn, cell_offn, 0, buf0);
      }
      if (mvalid == 4) kloop(cur, half, std::integral_constant<int, 4>()); else kloop(cur, half, std::integral_constant<int, 0>());
      GLDS_T(tc_);
      GLDS_C(cc_);
      GLDS_ACC(0, ta_, tb_); GLDS_ACC(1, tb_, tc_); GLDS_ACC(4, cb_, cc_);
    }
    GLDS_T(t4_);
    __syncthreads();            // every wave is done reading buf1: its first 16.6 KB become the four waves' transposition slabs
    GLDS_T(t5_);
    {
      float* R = resp + cell_off * nf;
      float* tr = (float*)(PERSIST ? buf1 : buf0) + wave * (16 * 65);           // per-wave [16 filters][64 cells + 1]
      const int pc = 16 * (wave + 4 * (lane >> 4)) + (lane & 15);
      const int pcy = pc / vw, py = y0 + pcy, pxx = x0 + (pc - pcy * vw);
      const bool pvalid = pc < ncell;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) tr[ai * 65 + m * 16 + 4 * ak + r] = acc[nt][m][r];   // D[i = 4 ak + r][j = ai] of M-tile m
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int jf = 0; jf < 16; ++jf) {
          const int fn = nbase + 16 * nt + jf;
          if (fn < nf && pvalid) R[(size_t)fn * H * W + (size_t)py * W + pxx] = tr[jf * 65 + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    GLDS_T(t6_);
    GLDS_ACC(2, t4_, t5_); GLDS_ACC(5, t5_, t6_);
    GLDS_ACC(6, 0ull, 1ull); GLDS_ACC(7, tl0 * 0ull + t0_, t6_);
    v = vn;
    y0 = y0n; x0 = x0n; W = Wn; H = Hn; cell_off = cell_offn;
  }
}

void launch_conv_glds_f32(const ConvTile* tiles, int ntiles, const LevelDev* levels, const float* feat, const float* wT,
                          float* resp, int nf, int nfpad, const float* border, int wg_per_cu, int ncu, hipStream_t s) {
  if (ntiles <= 0) return;
  const size_t lds = 2 * (size_t)(CT + 4) * (CT + 4) * 64;
  const int nwx = std::max(1, ncu / 8) * wg_per_cu;          // workgroups per XCD
  if (wg_per_cu <= 0) {    // one unit per workgroup, single buffer
    const int np = (nf + 31) / 32;
    static LdsOptIn optin;
    optin.ensure((const void*)k_conv_glds<3, false>, lds / 2);
    hipLaunchKernelGGL((k_conv_glds<3, false>), dim3((ntiles + 7) / 8 * 8 * np), dim3(256), lds / 2, s, tiles, levels, feat, wT, resp, nf, nfpad, ntiles, border);
  } else if (wg_per_cu >= 3) {
    static LdsOptIn optin;
    optin.ensure((const void*)k_conv_glds<3>, lds);
    hipLaunchKernelGGL((k_conv_glds<3>), dim3(8 * nwx), dim3(256), lds, s, tiles, levels, feat, wT, resp, nf, nfpad, ntiles, border);
  } else {
    static LdsOptIn optin;
    optin.ensure((const void*)k_conv_glds<2>, lds);
    hipLaunchKernelGGL((k_conv_glds<2>), dim3(8 * nwx), dim3(256), lds, s, tiles, levels, feat, wT, resp, nf, nfpad, ntiles, border);
  }
}

void launch_conv_mfma_f64(const ConvTile* tiles, int ntiles, const LevelDev* levels, const double* feat,
                          const double* wT, const double* w4u, double* resp, int nf, int nfpad, int kh, int kw, hipStream_t s) {
  if (ntiles <= 0) return;
  if (kh != 5 || kw != 5) {   // any other filter size: the same kernel with a run-time tap loop (16-byte B loads from the [tap][group][k][n][u] copy)
    launch_conv_mfma16_t<double, 4, 2, 1, true, 0, 0>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s, kh, kw);
    return;
  }
  // four 8-channel passes (27 KB of LDS per workgroup) measured 7 % faster than two 16-channel halves (54 KB)
  static const int q = PBD_PROBE_ENV("PBD_MFMA64_QUARTERS") ? atoi(PBD_PROBE_ENV("PBD_MFMA64_QUARTERS")) : 1;   // probe-build knob
  if (q == 2) launch_conv_mfma16_t<double, 4, 2>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);                // 8-byte B loads
  else if (q) launch_conv_mfma16_t<double, 4, 2, 1, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);      // 16-byte B loads (default)
  else launch_conv_mfma16_t<double, 2, 2>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);
}

// float instantiations of the same kernel: nhalf 3 = two channel halves, 3+ waves per SIMD (default fp32 filter
// bank: 0.39 ms and the best throughput with other frames' kernels co-resident), 2 = halves at 5 waves per SIMD
// (8 spilled registers), 1 = whole 32-channel tile in LDS (0.42 ms).  Tried and dropped: a persistent
// variant keeping the tile resident across a chunk of n-tiles with a register-direct epilogue (0.49 ms vs
// 0.44 ms, and long-running workgroups hurt the overlap with other frames' kernels); capping the kernel at
// two workgroups per CU to leave LDS and wave slots to co-running DT kernels (716 vs 751 frames/s); staging
// once for 2 or 5 n-tiles with a register-direct epilogue (one unaligned 16-byte store per M-tile and lane:
// 0.48-0.52 ms vs 0.43 ms — the LDS-transposed epilogue writes whole 64-byte row segments and is faster);
// double-buffered staging (next channel group prefetched into registers across the K loop, second LDS buffer):
// 0.51-0.71 ms vs 0.39 ms.
void launch_conv_mfma16_f32(const ConvTile* tiles, int ntiles, const LevelDev* levels, const float* feat,
                            const float* wT, const float* w4u, float* resp, int nf, int nfpad, int nhalf, hipStream_t s, int kh, int kw) {
  if (ntiles <= 0) return;
  if (kh != 5 || kw != 5) {   // any other filter size (3x3 .. 9x9): the default configuration (two n-tiles, 16-byte B loads) with a run-time tap loop
    launch_conv_mfma16_t<float, 2, 3, 2, true, 0, 0>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s, kh, kw);
    return;
  }
  if (nhalf == 20) launch_conv_mfma16_t<float, 2, 3, 2, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);   // two n-tiles per workgroup, 16-byte B loads
  else if (nhalf == 21) launch_conv_mfma16_t<float, 2, 3, 1, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);   // one n-tile, 16-byte B loads
  else if (nhalf == 22) launch_conv_mfma16_t<float, 2, 2, 2, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);   // two n-tiles, 2 waves/SIMD allocation
  else if (nhalf == 23) launch_conv_mfma16_t<float, 2, 2, 5, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);   // five n-tiles (80 filters), 16-byte B loads
  else if (nhalf == 24) launch_conv_mfma16_t<float, 2, 2, 3, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);   // three n-tiles (48 filters)
  else if (nhalf == 25) launch_conv_mfma16_t<float, 2, 4, 2, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);   // two n-tiles, register allocation for 4 waves per SIMD
  else if (nhalf == 26) launch_conv_mfma16_t<float, 2, 4, 1, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);   // one n-tile, 4 waves per SIMD
  else if (nhalf == 27) launch_conv_mfma16_t<float, 2, 5, 1, true>(tiles, ntiles, levels, feat, w4u, resp, nf, nfpad, s);   // one n-tile, 5 waves per SIMD
  else if (nhalf == 5) launch_conv_mfma16_t<float, 2, 3, 2>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);        // two n-tiles (32 filters) per workgroup
  else if (nhalf == 6) launch_conv_mfma16_t<float, 2, 2, 2>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);
  else if (nhalf == 7) launch_conv_mfma16_t<float, 1, 2, 2>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);   // whole tile, 32 filters
  else if (nhalf == 8) launch_conv_mfma16_t<float, 2, 2, 5>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);   // five n-tiles (80 filters) per workgroup
  else if (nhalf == 9) launch_conv_mfma16_t<float, 2, 3, 5>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);
  else if (nhalf == 2) launch_conv_mfma16_t<float, 2, 5>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);
  else if (nhalf == 3) launch_conv_mfma16_t<float, 2, 3>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);
  else if (nhalf == 4) launch_conv_mfma16_t<float, 4, 3>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);
  else launch_conv_mfma16_t<float, 1, 3>(tiles, ntiles, levels, feat, wT, resp, nf, nfpad, s);
}
