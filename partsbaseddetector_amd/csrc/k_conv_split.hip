// k_conv_split.hip — PBD_CONV_SPLIT: SpatialConvolutionEngine::pdf (reference src/SpatialConvolutionEngine.cpp:70-124,
// Filter2D src/filter.cpp:3879-3924) with the fp32 products carried by the bf16 matrix units through EXACT splits.
//
// An fp32 number is the sum of three bfloat16 (8 significant bits each): x = h + m + l, h = RN16(x), m = RN16(x - h),
// l = RN16(x - h - m), every subtraction exact.  A product of two bfloat16 is exact in an fp32 accumulator; the six partial
// products above 2^-24 relative (hh, hm, mh, hl, lh, mm) reproduce the fp32 product to fp32's own rounding
// (tests/tools_split_products_study.py: max error against fp64 3.1e-7 on the person bank, 9.1e-7 for the fp32 MFMA chain of
// k_conv_mfma16).  Why: fp32 MFMAs execute at the vector ALU's rate (157.3 TF is both peaks), so the bank and the distance
// transforms queue for one pipe; v_mfma_f32_32x32x16_bf16 runs on the matrix cores at 16x that rate.
//
// Implicit GEMM D[filter][cell] = sum over (tap, channel, product) — filters are the MFMA's A operand (rows), cells its B
// operand (columns): an accumulator register then holds 32 cells of ONE response plane, and a store writes 64-byte row segments
// without an LDS transpose.
//   * features: [cell][split][32 channels] bfloat16 in HBM (192 B per cell: k_feat_split, or k_hog's epilogue);
//   * filters:  [tap][k-step (16 channels)][split][32-filter n-tile][k-group (8 channels)][32 filters][8] bfloat16 — ONE
//     16-byte load per lane, k-step, split and n-tile, 1 KB contiguous per wavefront (host, once per model; L2-resident);
//   * a workgroup = NW wavefronts = a 16 x 4 NW cell unit of a ConvTile (default NW = 4: the whole tile); every wavefront owns two
//     32-cell M-tiles x NT (<= 5) 32-filter n-tiles = up to 160 accumulator registers, ONE wavefront per SIMD (the register file
//     is the occupancy bound): per k-step 15 filter loads + 6 LDS reads feed 60 MFMAs of 32 cycles — the operand traffic of a
//     64 x 160 register block is half the L1's rate where a 32 x 80 block saturates it (tests/tools/conv_split_probe.hip, round 4);
//   * the unit's halo tile sits in LDS as [split][cell][64 B], the four 16-byte channel groups of a cell XOR-swizzled with
//     bits 2-3 of the cell index: the 16 lanes of a ds_read_b128 lane group read 16 consecutive cells -> 16 different
//     16-byte bank slots (MI355X_MICROARCH.md, LDS: the lane groups are {0-3, 12-15, 20-27}, ...: the lane -> cell map of an
//     M-tile is permuted so that each group IS one row of 16 cells);
//   * the valid cells of a ragged unit are packed into M-tiles (level edges issue no MFMAs for cells that do not exist).
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>
#include <cstring>
#include "pbd_internal.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ static inline unsigned bf16_rn_bits(unsigned u) {   // fp32 bits -> bfloat16 bits, round to nearest even (finite, below bfloat16's overflow threshold)
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// ---- features -> three exact bfloat16 parts ([cell][split][32]); one thread = 8 consecutive channels of a cell ----
__global__ __launch_bounds__(256) void k_feat_split(const float* __restrict__ feat, uint16_t* __restrict__ out, size_t ngroups) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;    // (cell, channel group g = i & 3)
  if (i >= ngroups) return;
  const f32x4 a = *(const f32x4*)(feat + i * 8), b = *(const f32x4*)(feat + i * 8 + 4);
  float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  unsigned part[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float r = v[e];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      part[s][e] = bf16_rn_bits(__float_as_uint(r));
      r = r - __uint_as_float(part[s][e] << 16);               // exact: the difference has at most 16 (then 8) significant bits
    }
  }
  const size_t cell = i >> 2, g = i & 3;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    u32x4 w;
#pragma unroll
    for (int d = 0; d < 4; ++d) w[d] = part[s][2 * d] | (part[s][2 * d + 1] << 16);
    *(u32x4*)(out + (cell * 3 + s) * PBD_FLEN + g * 8) = w;
  }
}
void launch_feat_split(const float* feat, uint16_t* out, size_t ncells, hipStream_t s) {
  const size_t ngroups = ncells * 4;
  if (!ngroups) return;
  hipLaunchKernelGGL(k_feat_split, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s, feat, out, ngroups);
}

// ---- filters -> [tap][k-step][split][n-tile][k-group][32][8] bfloat16 (host) ----
int conv_split_ntiles(int nf) { return (nf + 31) / 32; }
void conv_split_filters(const float* filters, int nf, int kh, int kw, std::vector<uint16_t>& out) {
  const int ntap = kh * kw, NTL = conv_split_ntiles(nf);
  out.assign((size_t)ntap * 2 * 3 * NTL * 512, 0);
  for (int fn = 0; fn < nf; ++fn)
    for (int tap = 0; tap < ntap; ++tap)
      for (int c = 0; c < PBD_FLEN; ++c) {
        float r = filters[((size_t)fn * ntap + tap) * PBD_FLEN + c];
        const int ks = c >> 4, kg = (c >> 3) & 1, e = c & 7;
        for (int s = 0; s < 3; ++s) {
          unsigned u; memcpy(&u, &r, 4);
          const unsigned hb = bf16_rn_bits(u);
          out[((((size_t)(tap * 2 + ks) * 3 + s) * NTL + fn / 32) * 2 + kg) * 256 + (size_t)(fn % 32) * 8 + e] = (uint16_t)hb;
          const unsigned back = hb << 16; float hf; memcpy(&hf, &back, 4);
          r -= hf;                                              // exact
        }
      }
}

// NT: 32-filter n-tiles per workgroup (1..5); NW: wavefronts per workgroup (2: a 16 x 8 half of a ConvTile, 4: the whole 16 x 16 tile)
// PIN: the K loop's schedule for ONE wavefront per SIMD (see the loop): 1 = the next k-step's loads as a block in front of this k-step's MFMAs,
// 2 = the same loads dealt out between the MFMAs (sched_group_barrier); 0: hipcc's own schedule at two wavefronts per SIMD
// ---------------------------------------------------------------------------------------------------------------------
// PBD_CONV_SPLIT_F16 (opt-in): TWO binary16 parts per operand and THREE products.  binary16 carries 11 significant bits: with
// x 2^e = h + m (h = RN16(x 2^e), m = RN16(x 2^e - h), the subtraction exact) two parts hold 22-23 of an fp32 number's 24 bits,
// every product of two parts is exact in an fp32 accumulator, and h h + h m + m h leaves out terms of 2^-22 relative — against
// 2^-24 for the six bfloat16 products, but far under what the fp32 ACCUMULATION of 800 terms loses either way: measured against
// fp64 on the person bank (tests/tools_split_products_study.py) max 3.2e-7 / rms 4.1e-8, the six-product bank 3.1e-7 / 3.2e-8,
// the fp32 MFMA chain 9.1e-7 / 8.2e-8.  Half the matrix instructions of the six-product bank.
// binary16's range is the price: operands are scaled by powers of two (exact) to sit high in it — features by 2^12 (HOG features
// are <= 1: the truncation channel; |feature| must stay below 16), every filter's weights by its own 2^e with max |w| 2^e in
// [2^13, 2^14) — and a part below 2^-14 (scaled) is a binary16 subnormal of absolute precision 2^-25: an absolute error of 2^-37
// per feature, 2^-38 of the filter's max |w| per weight.  A filter's responses are multiplied by 2^-(12 + e) on the way out (exact).  Not the default and not what
// PBD_CONV_AUTO resolves to: the operands are represented to 23 bits, not 24 — bench.py reports it beside the benched bank.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SPLIT16_FEXP = 12;
__global__ __launch_bounds__(256) void k_feat_split16(const float* __restrict__ feat, uint16_t* __restrict__ out, size_t ngroups) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;    // (cell, channel group g = i & 3)
  if (i >= ngroups) return;
  const f32x4 a = *(const f32x4*)(feat + i * 8), b = *(const f32x4*)(feat + i * 8 + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  f16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = v[e] * (float)(1 << SPLIT16_FEXP);
    hi[e] = (_Float16)x;                                        // round to nearest even
    lo[e] = (_Float16)(x - (float)hi[e]);                       // (the subtraction is exact)
  }
  const size_t cell = i >> 2, g = i & 3;
  *(f16x8*)(out + (cell * 2 + 0) * PBD_FLEN + g * 8) = hi;
  *(f16x8*)(out + (cell * 2 + 1) * PBD_FLEN + g * 8) = lo;
}
void launch_feat_split16(const float* feat, uint16_t* out, size_t ncells, hipStream_t s) {
  const size_t ngroups = ncells * 4;
  if (!ngroups) return;
  hipLaunchKernelGGL(k_feat_split16, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s, feat, out, ngroups);
}
// filters -> [tap][k-step][part (2)][n-tile][k-group][32][8] binary16 of w 2^e(filter); oscale[filter] = the response scale 2^-(12 + e)
void conv_split16_filters(const float* filters, int nf, int kh, int kw, std::vector<uint16_t>& out, std::vector<float>& oscale) {
  const int ntap = kh * kw, NTL = conv_split_ntiles(nf);
  out.assign((size_t)ntap * 2 * 2 * NTL * 512, 0);
  oscale.assign((size_t)NTL * 32, 1.f);
  for (int fn = 0; fn < nf; ++fn) {
    const float* wf = filters + (size_t)fn * ntap * PBD_FLEN;
    float wmax = 0.f;
    for (int i = 0; i < ntap * PBD_FLEN; ++i) wmax = std::max(wmax, std::fabs(wf[i]));
    int x = 0;
    if (wmax > 0.f) std::frexp(wmax, &x);                     // wmax = f 2^x, f in [0.5, 1)
    const int e = wmax > 0.f ? std::min(100, std::max(-110, 14 - x)) : 0;
    oscale[fn] = std::ldexp(1.0f, -(SPLIT16_FEXP + e));
    for (int tap = 0; tap < ntap; ++tap)
      for (int c = 0; c < PBD_FLEN; ++c) {
        float r = std::ldexp(filters[((size_t)fn * ntap + tap) * PBD_FLEN + c], e);
        const int ks = c >> 4, kg = (c >> 3) & 1, el = c & 7;
        for (int s = 0; s < 2; ++s) {
          const _Float16 hv = (_Float16)r;
          uint16_t bits; memcpy(&bits, &hv, 2);
          out[((((size_t)(tap * 2 + ks) * 2 + s) * NTL + fn / 32) * 2 + kg) * 256 + (size_t)(fn % 32) * 8 + el] = bits;
          r -= (float)hv;                                       // exact
        }
      }
  }
}

// NS: parts per operand — 3: bfloat16 parts, six products (PBD_CONV_SPLIT); 2: scaled binary16 parts, three products (PBD_CONV_SPLIT_F16, below)
template <int NT, int NW, int PIN, int NS = 3>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, PIN ? 1 : 2))) void k_conv_split32(const ConvTile* __restrict__ tiles, const LevelDev* __restrict__ levels,
                                                             const uint16_t* __restrict__ feat, const uint16_t* __restrict__ filt,
                                                             float* __restrict__ resp, int nf, int ntl_bank, int ntile0, int ngroups,
                                                             int ntiles_total, int kh, int kw, const float* __restrict__ oscale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using opnd = std::conditional_t<NS == 3, bf16x8, f16x8>;
  constexpr int ROWS = 4 * NW, NHALVES = 16 / ROWS, NTHR = 64 * NW, PPC = 4 * NS;   // PPC: 16-byte pieces per cell
  const int TW = 16 + kw - 1, TH = ROWS + kh - 1, NC = TW * TH, PLANE = NC * 64;
  // workgroup -> (tile, role = (half, n-group)): tile position 8 g + x runs on XCD x (the plan pairs horizontal neighbours on that
  // convention); the roles of a tile share lin % 8 (one XCD: the halves' common halo rows and the n-groups' common tile come from
  // HBM once) and are dispatched 8 workgroups apart
  const int R = NHALVES * ngroups;
  const int lin = blockIdx.x;
  const int grp = lin / (8 * R), rem = lin - grp * (8 * R);
  const int tile_i = grp * 8 + (rem & 7), role = rem >> 3;
  if (tile_i >= ntiles_total) return;
  const int half = role & (NHALVES - 1), ngroup = role / NHALVES;
  const ConvTile t = tiles[tile_i];
  const LevelDev lv = levels[t.level];
  const int H = lv.ch, W = lv.cw;
  const int ty0 = t.y0 + ROWS * half, tx0 = t.x0;
  if (ty0 >= H) return;                                  // the lower half of a tile on the level's last rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint16_t* F = feat + lv.cell_off * (NS * PBD_FLEN);
  {  // stage the halo tile: 4 NS 16-byte pieces per cell (piece = 4 split + channel group), batches of independent loads; outside the
     // level: zeros, and 1.0 (0x3F80, exact in bfloat16: part h; binary16 parts: 2^12 = 0x6C00) in the truncation channel = element 7 of piece 3 (:147-155)
    const int NPC = NC * PPC, oy = ty0 - kh / 2, ox = tx0 - kw / 2;
    constexpr unsigned ONE = NS == 3 ? 0x3F800000u : 0x6C000000u;
    auto cell_of = [](int i) { return NS == 3 ? (int)(((unsigned)i * 43691u) >> 19) : i >> 3; };     // i / 12 (exact for i < 2^17), i / 8
    const unsigned magic_tw = 0xFFFFFFFFu / (unsigned)TW + 1u;     // cell / TW = umulhi(cell, magic) (cell * TW < 2^32)
    constexpr int BATCH = 12;
    for (int i0 = 0; i0 < NPC; i0 += NTHR * BATCH) {
      u32x4 v[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int i = min(i0 + j * NTHR + tid, NPC - 1);
        const int cell = cell_of(i), piece = i - cell * PPC;
        const int cy = (int)__umulhi((unsigned)cell, magic_tw), cx = cell - cy * TW;
        const int y = min(max(oy + cy, 0), H - 1), x = min(max(ox + cx, 0), W - 1);
        v[j] = *(const u32x4*)(F + ((size_t)(y * W + x) * (NS * PBD_FLEN) + piece * 8));
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int i = i0 + j * NTHR + tid;
        if (i < NPC) {
          const int cell = cell_of(i), piece = i - cell * PPC;
          const int cy = (int)__umulhi((unsigned)cell, magic_tw), cx = cell - cy * TW;
          const int y = oy + cy, x = ox + cx;
          const bool inside = y >= 0 && y < H && x >= 0 && x < W;
          const u32x4 border = u32x4{0u, 0u, 0u, piece == 3 ? ONE : 0u};
          const int q = piece & 3;
          *(u32x4*)(smem + (piece >> 2) * PLANE + cell * 64 + ((q ^ ((cell >> 2) & 3)) << 4)) = inside ? v[j] : border;
        }
      }
    }
  }
  __syncthreads();
  // lane -> (cell position inside a 32-cell M-tile, k-group).  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27},
  // {4-11, 16-19, 28-31} (+32): positions are dealt so that each group is 16 consecutive positions = one row of a full-width unit
  const int c = lane & 31, kg = lane >> 5;
  const int mid = ((c & 15) >= 4 && (c & 15) < 12) ? 1 : 0;
  const int pos = (c & 15) + 16 * (mid ^ (c >> 4));
  // packed M-tiles: valid cell number 32 j + pos of the vh x vw valid region, j = wave + NW m (round robin: a ragged unit's
  // M-tiles spread over the wavefronts); positions past the last cell repeat it (never stored)
  const int vw = min(16, W - tx0), vh = min(ROWS, H - ty0), ncell = vw * vh;
  const int nmt = (ncell + 31) >> 5;
  const int mvalid = __builtin_amdgcn_readfirstlane(max(0, min(2, (nmt - wave + NW - 1) / NW)));
  const unsigned vw_magic = 65535u / (unsigned)vw + 1u;   // idx / vw for idx < 256 (k_conv_mfma16)
  int cl0[2], cofs[2];
  bool cval[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int idx = 32 * (wave + NW * m) + pos;
    cval[m] = idx < ncell;
    const int ic = min(idx, ncell - 1);
    const int cy = (int)(((unsigned)ic * vw_magic) >> 16), cx = ic - cy * vw;
    cl0[m] = cy * TW + cx;
    cofs[m] = (ty0 + cy) * W + tx0 + cx;
  }
  const int ntb = ntile0 + ngroup * NT;                  // first n-tile of this workgroup
  f32x16 acc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][m][r] = 0.f;
  const uint16_t* bl = filt + (size_t)ntb * 512 + lane * 8;
  const size_t bs_split = (size_t)ntl_bank * 512, bs_kstep = NS * bs_split;
  const int nkstep = 2 * kh * kw;
#ifdef PBD_BANK_PRIO   // experiment builds only: the K loop's wavefronts at a raised issue priority against the other batches' kernels on the SIMD
  __builtin_amdgcn_s_setprio(PBD_BANK_PRIO);
#endif

  auto k_loop = [&](auto mv_tag) __attribute__((always_inline)) {
    constexpr int MV = decltype(mv_tag)::value;
    auto load_b = [&](opnd (&b)[NT][NS], int kstep) __attribute__((always_inline)) {
      const uint16_t* p = bl + (size_t)min(kstep, nkstep - 1) * bs_kstep;
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt][s] = *(const opnd*)(p + s * bs_split + nt * 512);
    };
    auto load_a = [&](opnd (&a)[2][NS], int tapofs, int ks) __attribute__((always_inline)) {     // tapofs = ti * TW + tj
#pragma unroll
      for (int m = 0; m < MV; ++m) {
        const int cl = cl0[m] + tapofs;
        const char* p = smem + cl * 64 + (((2 * ks + kg) ^ ((cl >> 2) & 3)) << 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) a[m][s] = *(const opnd*)(p + s * PLANE);
      }
    };
    auto mma = [&](const opnd (&a)[2][NS], const opnd (&b)[NT][NS]) __attribute__((always_inline)) {
      // products outermost (consecutive MFMAs go to different accumulators: an accumulator is touched every 2 NT instructions)
      auto sweep = [&](int sa, int sb) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int m = 0; m < MV; ++m) {
            if constexpr (NS == 3) acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[nt][sb], a[m][sa], acc[nt][m], 0, 0, 0);
            else acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[nt][sb], a[m][sa], acc[nt][m], 0, 0, 0);
          }
      };
      // feature part x filter part, the small products of a k-step before its large one: m m, h l, l h (2^-16), h m, m h (2^-8), h h
      // (binary16 parts: h m, m h (2^-11), h h)
      if constexpr (NS == 3) { sweep(1, 1); sweep(0, 2); sweep(2, 0); }
      sweep(0, 1); sweep(1, 0); sweep(0, 0);
    };
    int ti = 0, tj = 0;
    const int ntap = kh * kw;
    // (binary16 parts: a k-step is 30 MFMAs = 960 cycles, and SQ_WAIT_INST_ANY reads 32 % of the wavefronts' cycles.  The filters TWO k-steps
    //  ahead — three register sets in rotation, a body of three taps — measured the same pdf time, 0.141 vs 0.139-0.141 ms per frame, at 430
    //  registers: no distance-transform wavefront beside it, 2 150 vs 2 225-2 277 frames/s; r05 session 15.  Not latency: the operand traffic.)
    opnd a0[2][NS], a1[2][NS], b0[NT][NS], b1[NT][NS];
    load_b(b0, 0);
    load_a(a0, 0, 0);
#pragma unroll 1
    for (int tap = 0; tap < ntap; ++tap) {
      // operands in explicit ping-pong, the schedule pinned: the next k-step's 15 filter loads + 6 LDS reads are ISSUED before this
      // k-step's 60 MFMAs (1 920 cycles) and waited for after them.  Left alone, hipcc's scheduler sinks every load to just in front
      // of its first use to save registers (80 VGPRs) — an L2 round trip in front of every other MFMA, with one wavefront per SIMD
      const int tapofs = ti * TW + tj;
      auto deal = [&]() {   // PIN == 2: one load per three MFMAs, then one LDS read per two (the MFMA pipe never waits for an issue burst)
        if constexpr (PIN == 2) {
#pragma unroll
          for (int i = 0; i < NS * NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, NS == 3 ? (MV == 2 ? 3 : 1) : MV, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
#pragma unroll
          for (int i = 0; i < NS * MV; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        }
      };
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      load_b(b1, 2 * tap + 1);
      load_a(a1, tapofs, 1);
      if (PIN == 1) __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      deal();
      if (PIN) __builtin_amdgcn_sched_barrier(0);
      if (++tj == kw) { tj = 0; ++ti; }
      const int nextofs = tap + 1 < ntap ? ti * TW + tj : tapofs;   // (past the last tap: this tap again, never used)
      load_b(b0, 2 * tap + 2);
      load_a(a0, nextofs, 0);
      if (PIN == 1) __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      deal();
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (mvalid == 2) k_loop(std::integral_constant<int, 2>());
  else if (mvalid == 1) k_loop(std::integral_constant<int, 1>());
  if (mvalid == 0) return;

  // D[i = filter 32 nt + (r & 3) + 8 (r >> 2) + 4 kg][j = cell pos of M-tile m]
  float* Rl = resp + lv.cell_off * nf;
  const size_t HW = (size_t)H * W;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    if (m < mvalid && cval[m]) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float* pl = Rl + (size_t)(32 * (ntb + nt) + 4 * kg) * HW + cofs[m];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int fo = (r & 3) + 8 * (r >> 2);
          if (32 * (ntb + nt) + 4 * kg + fo < nf) pl[(size_t)fo * HW] = NS == 3 ? acc[nt][m][r] : acc[nt][m][r] * oscale[32 * (ntb + nt) + 4 * kg + fo];   // (a power of two: exact)
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_conv_split32p — the 5 x 5 bank as a PERSISTENT workgroup with the tile staging hidden under the MFMAs.  In k_conv_split32
// the matrix pipe is busy 58-64 % of the kernel's time (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, profiles/r05*): with one
// wavefront per SIMD nothing runs while a workgroup stages its tile (global -> registers -> LDS, a memory round trip per batch of
// loads) or stores its 160 x 256 results.  Here a workgroup (4 wavefronts, one per CU by registers) walks a list of units (tile,
// n-group); the K loop runs over the two 16-channel halves of the tile one after the other — (half 0: taps 0..24), (half 1: taps
// 0..24) — and while one half (38 KB: [split][cell][32 B]) feeds the MFMAs, the next half — of this unit, or half 0 of the next
// unit — is fetched piece by piece between the MFMAs into the other buffer: the same 77 KB of LDS as the one-shot kernel (a
// distance-transform block of another batch still fits beside it), two barriers per unit, no staging phase.  The filter loads run
// on across the halves and units (the last k-step of a half prefetches the first of the next).
// Units: tile positions x, x + 8, ... of the plan's list belong to XCD x = blockIdx.x % 8 (k_conv_split32's convention); workgroup
// j of the XCD takes its units j, j + nwx, ...
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void conv_split32p_body(char* smem, const ConvTile* __restrict__ tiles, const LevelDev* __restrict__ levels,
                                                   const uint16_t* __restrict__ feat, const uint16_t* __restrict__ filt,
                                                   float* __restrict__ resp, int nf, int ntl_bank, int ntile0, int ngroups,
                                                   int ntiles_total) {
  constexpr int NW = 4, KW = 5, TW = 20, NC = 400, HPLANE = NC * 32, HBUF = 3 * HPLANE, NTAP = 25, NHP = NC * 6;   // NHP: 16-byte pieces of a half tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, nwx = gridDim.x >> 3;
  const int ntx = ntiles_total > xcd ? (ntiles_total - xcd + 7) >> 3 : 0;
  const int nunits = ntx * ngroups;
  if (j0 >= nunits) return;
  const int c = lane & 31, kg = lane >> 5;
  const int mid = ((c & 15) >= 4 && (c & 15) < 12) ? 1 : 0;
  const int pos = (c & 15) + 16 * (mid ^ (c >> 4));          // (k_conv_split32: each ds_read_b128 lane group = 16 consecutive cells)
  const size_t bs_split = (size_t)ntl_bank * 512, bs_kstep = 3 * bs_split;

  struct Unit { int y0, x0, W, H, ng; size_t cell_off; };
  auto unit_of = [&](int v) {
    const int ti = xcd + 8 * (v / ngroups);
    const ConvTile t = tiles[ti];
    const LevelDev lv = levels[t.level];
    return Unit{t.y0, t.x0, lv.cw, lv.ch, v - (v / ngroups) * ngroups, (size_t)lv.cell_off};
  };
  // piece jj of this thread of half `half` of unit U's tile: the global address (clamped) ...
  auto piece_src = [&](const Unit& U, int half, int jj) {
    const int i = min(jj * 256 + tid, NHP - 1);
    const int cell = (int)(((unsigned)i * 43691u) >> 18), p6 = i - cell * 6;          // i / 6 (exact for i < 2^16)
    const int cy = (int)(((unsigned)cell * 52429u) >> 20), cx = cell - cy * TW;        // cell / 20
    const int y = min(max(U.y0 - 2 + cy, 0), U.H - 1), x = min(max(U.x0 - 2 + cx, 0), U.W - 1);
    const int gp = (p6 >> 1) * 4 + 2 * half + (p6 & 1);                                // piece of the cell's 192 bytes: 4 split + channel group
    return (const u32x4*)(feat + (U.cell_off + (size_t)(y * U.W + x)) * (3 * PBD_FLEN) + gp * 8);
  };
  // ... and where it goes: border value (0, and 1.0 = 0x3F80 in the truncation channel: half 1, part h, second group, element 7) if the cell
  // lies outside the level; [split][cell][32 B] with the two 16-byte groups of a cell swapped in every other run of 8 cells (16 consecutive
  // cells x 16 bytes then cover all 16 bank slots)
  auto piece_put = [&](const Unit& U, int half, int jj, char* buf, u32x4 v) {
    const int i = min(jj * 256 + tid, NHP - 1);
    const int cell = (int)(((unsigned)i * 43691u) >> 18), p6 = i - cell * 6;
    const int cy = (int)(((unsigned)cell * 52429u) >> 20), cx = cell - cy * TW;
    const int y = U.y0 - 2 + cy, x = U.x0 - 2 + cx;
    // (a mask, not a select: hipcc turns `inside ? v : border` into a conditional LOAD at the point of use — the fetch issued a k-step early
    //  is then gone, and the branch cuts the k-step's scheduling region in two)
    const unsigned m = 0u - (unsigned)((int)(y >= 0) & (int)(y < U.H) & (int)(x >= 0) & (int)(x < U.W));
    const unsigned b3 = (half == 1 && p6 == 1) ? 0x3F800000u : 0u;
    const u32x4 w = u32x4{v[0] & m, v[1] & m, v[2] & m, (v[3] & m) | (b3 & ~m)};
    *(u32x4*)(buf + (p6 >> 1) * HPLANE + cell * 32 + (((p6 & 1) ^ ((cell >> 3) & 1)) << 4)) = w;
  };
  constexpr int NPT = (NHP + 255) / 256;                     // pieces per thread and half tile (10)
  static_assert(2 * NPT <= NTAP - 1, "a half tile is staged under the 25 k-steps of the previous half");

  int v = j0;
  Unit U = unit_of(v);
  {  // the first unit's first half: nothing to hide it under
    u32x4 r[NPT];
#pragma unroll
    for (int jj = 0; jj < NPT; ++jj) r[jj] = *piece_src(U, 0, jj);
#pragma unroll
    for (int jj = 0; jj < NPT; ++jj) piece_put(U, 0, jj, smem, r[jj]);
  }
  const uint16_t* bl = filt + (size_t)(ntile0 + U.ng * NT) * 512 + lane * 8;
  bf16x8 a0[2][3], a1[2][3], b0[NT][3], b1[NT][3];
  auto load_b = [&](bf16x8 (&b)[NT][3], const uint16_t* p) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt][s] = *(const bf16x8*)(p + s * bs_split + nt * 512);
  };
  load_b(b0, bl);                                            // k-step (tap 0, half 0) of the first unit
  f32x16 acc[NT][2];

  while (true) {
    const int vn = v + nwx;
    const bool has_next = vn < nunits;
    const Unit Un = unit_of(has_next ? vn : v);              // (after the last unit: this unit again — its half 0 is staged once more, unused)
    const uint16_t* bln = filt + (size_t)(ntile0 + Un.ng * NT) * 512 + lane * 8;
    // packed M-tiles of the unit's valid cells (k_conv_split32)
    const int vw = min(16, U.W - U.x0), vh = min(16, U.H - U.y0), ncell = vw * vh;
    const int nmt = (ncell + 31) >> 5;
    const int mvalid = __builtin_amdgcn_readfirstlane(max(0, min(2, (nmt - wave + NW - 1) / NW)));
    const unsigned vw_magic = 65535u / (unsigned)vw + 1u;
    int cl0[2], cofs[2];
    bool cval[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int idx = 32 * (wave + NW * m) + pos;
      cval[m] = idx < ncell;
      const int ic = min(idx, ncell - 1);
      const int cy = (int)(((unsigned)ic * vw_magic) >> 16), cx = ic - cy * vw;
      cl0[m] = cy * TW + cx;
      cofs[m] = (U.y0 + cy) * U.W + U.x0 + cx;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][m][r] = 0.f;

    // one half of the unit: 25 k-steps (one per tap) on buffer `bufc`, operands in ping-pong (ac / bc hold the k-step being multiplied,
    // an / bn receive the next one's), one staging action per k-step: even k-steps fetch a piece of the half staged next (unit Us, half hs),
    // odd ones write it to `bufn`.  bnext: the filters of the k-step after this half's last.
    auto run_half = [&](auto mv_tag, int half, const char* bufc, char* bufn, const Unit& Us, int hs, const uint16_t* bnext,
                        bf16x8 (&ac)[2][3], bf16x8 (&bc)[NT][3], bf16x8 (&an)[2][3], bf16x8 (&bn)[NT][3]) __attribute__((always_inline)) {
      constexpr int MV = decltype(mv_tag)::value;
      auto load_a = [&](bf16x8 (&a)[2][3], int tapofs) {
#pragma unroll
        for (int m = 0; m < (MV ? MV : 1); ++m) {
          if (MV == 0) break;
          const int cl = cl0[m] + tapofs;
          const char* p = bufc + cl * 32 + ((kg ^ ((cl >> 3) & 1)) << 4);
#pragma unroll
          for (int s = 0; s < 3; ++s) a[m][s] = *(const bf16x8*)(p + s * HPLANE);
        }
      };
      auto mma = [&](const bf16x8 (&a)[2][3], const bf16x8 (&b)[NT][3]) {
        auto sweep = [&](int sa, int sb) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int m = 0; m < MV; ++m) acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[nt][sb], a[m][sa], acc[nt][m], 0, 0, 0);
        };
        sweep(1, 1); sweep(0, 2); sweep(2, 0); sweep(0, 1); sweep(1, 0); sweep(0, 0);
      };
      // the next k-step's loads dealt out between this k-step's MFMAs (k_conv_split32, PIN == 2); extra: the block's staging action
      auto deal = [&](int extra_vmem, int extra_dsw) {
        if constexpr (MV > 0) {
#pragma unroll
          for (int i = 0; i < 3 * NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, MV == 2 ? 3 : 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
          if (extra_vmem) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); }
          if (extra_dsw) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#pragma unroll
          for (int i = 0; i < 3 * MV; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, MV == 2 ? 2 : 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        }
      };
      const uint16_t* bh = bl + (size_t)half * bs_kstep;       // k-step (tap, half) = filters' k-step 2 tap + half
      __syncthreads();                                          // `bufc` is complete (its pieces were written under the previous half); nobody reads `bufn` any more
      load_a(ac, 0);
      u32x4 pv = u32x4{0u, 0u, 0u, 0u};
      int ti = 0, tj = 0;
#pragma unroll 1
      for (int tp = 0; tp < NTAP / 2; ++tp) {                   // taps 2 tp, 2 tp + 1
        __builtin_amdgcn_sched_barrier(0);
        pv = *piece_src(Us, hs, min(tp, NPT - 1));
        if (MV) load_b(bn, bh + (size_t)(2 * (2 * tp + 1)) * bs_kstep);
        if (++tj == KW) { tj = 0; ++ti; }
        load_a(an, ti * TW + tj);
        mma(ac, bc);
        deal(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        piece_put(Us, hs, min(tp, NPT - 1), bufn, pv);           // (taps past the half's last piece write that piece once more)
        if (MV) load_b(bc, bh + (size_t)(2 * (2 * tp + 2)) * bs_kstep);
        if (++tj == KW) { tj = 0; ++ti; }
        load_a(ac, ti * TW + tj);
        mma(an, bn);
        deal(0, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      // tap 24: the operands are in (ac, bc); the first k-step of the next half goes to bn (its A operands are read after the barrier)
      __builtin_amdgcn_sched_barrier(0);
      if (MV) load_b(bn, bnext);
      mma(ac, bc);
      deal(0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto run_unit = [&](auto mv_tag) __attribute__((always_inline)) {
      // half 0 on buffer 0 (staging this unit's half 1 into buffer 1), then half 1 on buffer 1 (staging the next unit's half 0 into buffer 0);
      // 25 k-steps per half: the ping-pong roles swap from half to half and are back after the unit
      run_half(mv_tag, 0, smem, smem + HBUF, U, 1, bl + bs_kstep, a0, b0, a1, b1);
      run_half(mv_tag, 1, smem + HBUF, smem, Un, 0, bln, a1, b1, a0, b0);
    };
    // (every wavefront runs both of its M-tiles, valid or not — a tile past the unit's last cell repeats that cell and is never stored: the
    //  workgroup meets at a barrier per half anyway, so a wavefront that skipped MFMAs would only wait there, and ONE instantiation of the
    //  K loop keeps the kernel at 406 registers instead of 478: a distance-transform wavefront of another batch still fits on the SIMD)
    run_unit(std::integral_constant<int, 2>());

    // D[i = filter 32 nt + (r & 3) + 8 (r >> 2) + 4 kg][j = cell pos of M-tile m]
    {
      float* Rl = resp + U.cell_off * nf;
      const size_t HW = (size_t)U.H * U.W;
      const int ntb = ntile0 + U.ng * NT;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m < mvalid && cval[m]) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float* pl = Rl + (size_t)(32 * (ntb + nt) + 4 * kg) * HW + cofs[m];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int fo = (r & 3) + 8 * (r >> 2);
              if (32 * (ntb + nt) + 4 * kg + fo < nf) pl[(size_t)fo * HW] = acc[nt][m][r];
            }
          }
        }
      }
    }
    if (!has_next) break;
    v = vn; U = Un; bl = bln;
  }
}

template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_conv_split32p(const ConvTile* __restrict__ tiles, const LevelDev* __restrict__ levels,
    const uint16_t* __restrict__ feat, const uint16_t* __restrict__ filt, float* __restrict__ resp, int nf, int ntl_bank, int ntile0, int ngroups, int ntiles_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_split32p_body<NT>(smem, tiles, levels, feat, filt, resp, nf, ntl_bank, ntile0, ngroups, ntiles_total);
}
template <int NT>
static void launch_conv_split_p(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                                float* resp, int nf, int ntl_bank, int ntile0, int ngroups, int ncu, hipStream_t s) {
  const size_t lds = 2 * 3 * 400 * 32;
  static LdsOptIn optin;
  optin.ensure((const void*)k_conv_split32p<NT>, lds);
  const int grid = 8 * std::max(1, (ncu + 7) / 8);          // one workgroup per CU (by registers), the same number on every XCD
  hipLaunchKernelGGL((k_conv_split32p<NT>), dim3(grid), dim3(256), lds, s, tiles, levels, feat_split, wS, resp, nf, ntl_bank, ntile0, ngroups, ntiles);
}
// 5 x 5 banks only (two half tiles of 20 x 20 cells in LDS).  TUNING BUILDS ONLY (PBD_SPLIT_VARIANT=6), measured and not adopted (r05 session 4,
// alternating runs): pdf 0.172-0.177 ms per frame in batches of 8 against 0.169 for k_conv_split32's default, 1 815-1 853 against 2 034-2 043 frames/s
// with three batches in flight (406 registers: no distance-transform wavefront fits beside it; capped at 400 with amdgpu_num_vgpr the K loop
// spills: 0.316 ms).  The one-shot kernel's staging phase is NOT what holds its matrix pipe at 58-64 %: at ~1.2 PF of bf16 products on random
// data the chip runs at its power limit (effective clock ~2.0 GHz; the guides' best HIP GEMM sustains 1.3 PF on random operands).
void launch_conv_split_persistent(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                                  float* resp, int nf, int ncu, hipStream_t s) {
  if (ntiles <= 0) return;
  const int ntl = conv_split_ntiles(nf), full = ntl / 5, rest = ntl - 5 * full;
  if (full) launch_conv_split_p<5>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, 0, full, ncu, s);
  switch (rest) {
    case 1: launch_conv_split_p<1>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, 5 * full, 1, ncu, s); break;
    case 2: launch_conv_split_p<2>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, 5 * full, 1, ncu, s); break;
    case 3: launch_conv_split_p<3>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, 5 * full, 1, ncu, s); break;
    case 4: launch_conv_split_p<4>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, 5 * full, 1, ncu, s); break;
    default: break;
  }
}

template <int NT, int NW, int PIN, int NS = 3>
static void launch_conv_split_t(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                                float* resp, int nf, int ntl_bank, int ntile0, int ngroups, int kh, int kw, hipStream_t s, const float* oscale = nullptr) {
  constexpr int ROWS = 4 * NW, NHALVES = 16 / ROWS;
  const size_t lds = (size_t)(16 + kw - 1) * (ROWS + kh - 1) * 64 * NS;
  static LdsOptIn optin;
  optin.ensure((const void*)k_conv_split32<NT, NW, PIN, NS>, lds);
  const int grid = (ntiles + 7) / 8 * 8 * NHALVES * ngroups;
  hipLaunchKernelGGL((k_conv_split32<NT, NW, PIN, NS>), dim3(grid), dim3(64 * NW), lds, s, tiles, levels, feat_split, wS, resp, nf, ntl_bank,
                     ntile0, ngroups, ntiles, kh, kw, oscale);
}
// PBD_CONV_SPLIT_F16: the default form of the six-product bank (4 wavefronts per workgroup, dealt loads, groups of five n-tiles) over two parts
// variant (tuning builds, PBD_SPLIT_VARIANT): 4 = hipcc's own schedule at two wavefronts per SIMD, 7 / 8 = groups of four / three n-tiles
template <int PIN>
static void launch_conv_split16_g(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                                  float* resp, int nf, int kh, int kw, const float* oscale, int G, hipStream_t s) {
  const int ntl = conv_split_ntiles(nf), full = ntl / G, rest = ntl - G * full;
  auto go = [&](int nt, int ntile0, int ngroups) {
    switch (nt) {
      case 1: launch_conv_split_t<1, 4, PIN, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s, oscale); break;
      case 2: launch_conv_split_t<2, 4, PIN, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s, oscale); break;
      case 3: launch_conv_split_t<3, 4, PIN, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s, oscale); break;
      case 4: launch_conv_split_t<4, 4, PIN, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s, oscale); break;
      case 5: launch_conv_split_t<5, 4, PIN, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s, oscale); break;
      default: break;
    }
  };
  if (full) go(G, 0, full);
  if (rest) go(rest, G * full, 1);
}
void launch_conv_split16(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                         float* resp, int nf, int kh, int kw, const float* oscale, int variant, hipStream_t s) {
  if (ntiles <= 0) return;
  if (variant == 4) launch_conv_split16_g<0>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, oscale, 5, s);
  else launch_conv_split16_g<2>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, oscale, variant == 7 ? 4 : variant == 8 ? 3 : 5, s);
}
template <int NW, int PIN>
static void launch_conv_split_nw(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                                 float* resp, int nf, int kh, int kw, hipStream_t s, int G = 5) {
  // groups of G = five n-tiles (160 filters: the person bank's 156 in one pass), then the remainder with its own instantiation
  // (G = 4 / 3: tuning variants — fewer accumulators per wavefront, two wavefronts per SIMD, every tile staged once per group)
  // (banks of more than 160 filters: balanced groups — 208 filters as 4 + 3 n-tiles instead of 5 + 2 — measured the same, 0.300 vs 0.302 ms per frame: what a
  //  second group costs is staging every tile again and a second launch tail, not the smaller register block; session 11)
  const int ntl = conv_split_ntiles(nf);
  const int full = ntl / G, rest = ntl - G * full;
  auto go = [&](int nt, int ntile0, int ngroups) {
    switch (nt) {
      case 1: launch_conv_split_t<1, NW, PIN>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s); break;
      case 2: launch_conv_split_t<2, NW, PIN>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s); break;
      case 3: launch_conv_split_t<3, NW, PIN>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s); break;
      case 4: launch_conv_split_t<4, NW, PIN>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s); break;
      case 5: launch_conv_split_t<5, NW, PIN>(tiles, ntiles, levels, feat_split, wS, resp, nf, ntl, ntile0, ngroups, kh, kw, s); break;
      default: break;
    }
  };
  if (full) go(G, 0, full);
  if (rest) go(rest, G * full, 1);
}
// Default (variant 0): four wavefronts per workgroup = one 16 x 16 cell ConvTile, ONE workgroup per CU (298 registers per
// wavefront), the next k-step's loads dealt out between this k-step's MFMAs.  Measured on the MI355X (profiles/r05*: pdf per frame
// in batches of 8 / whole-pipeline frames per second with three batches in flight): 0.168 ms / 2 090-2 116; the same with the loads
// issued as a block in front of the MFMAs 0.180 / 1 900; two-wavefront workgroups (16 x 8 cell units, two per CU) 0.174 / 1 830
// and 0.186 / 1 830; hipcc's own schedule (loads sunk to their uses, 240 registers, two wavefronts per SIMD) 0.187 / 1 960 with
// four wavefronts per workgroup, 0.225 / 1 685 with two.  One workgroup per CU leaves the CU's other LDS half and a quarter of its
// registers to a distance-transform block of another batch in flight, whose integer / fp64 vector work issues beside bf16 MFMAs
// (tests/tools/mfma_valu_overlap_probe.hip: 0.5-0.75 of the shorter one hidden; fp32 FMAs: none).
// variant (tuning builds, PBD_SPLIT_VARIANT): 1 = loads as a block, 2 / 3 = the two-wavefront forms of 0 / 1, 4 / 5 = hipcc's schedule (4 / 2 wavefronts)
void launch_conv_split(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                       float* resp, int nf, int kh, int kw, int variant, hipStream_t s) {
  if (ntiles <= 0) return;
  if (variant == 1) launch_conv_split_nw<4, 1>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, s);
  else if (variant == 2) launch_conv_split_nw<2, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, s);
  else if (variant == 3) launch_conv_split_nw<2, 1>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, s);
  else if (variant == 4) launch_conv_split_nw<4, 0>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, s);
  else if (variant == 5) launch_conv_split_nw<2, 0>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, s);
  else if (variant == 7) launch_conv_split_nw<4, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, s, 4);   // groups of four n-tiles (+ remainder)
  else if (variant == 8) launch_conv_split_nw<4, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, s, 3);   // groups of three
  else launch_conv_split_nw<4, 2>(tiles, ntiles, levels, feat_split, wS, resp, nf, kh, kw, s);
}
