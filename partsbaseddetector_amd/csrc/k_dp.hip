// k_dp.hip — DynamicProgram<T>::min / argmin on the GPU.
// Reference: src/DynamicProgram.cpp:66-255, include/DistanceTransform.hpp:151-245,
// include/Math.hpp:108-185.
//
// k_dt_pass   one 1-D generalised distance transform pass (Felzenszwalb &
//             Huttenlocher upper envelope, DistanceTransform.hpp:151-182) over all
//             lines of a round: a block = 64 / 128 / 256 lanes = the lines that fit its LDS
//             budget, several lanes per line, each scanning one segment of the line;
//             the segments are stitched into the result of the sequential run
//             (dt_core.hpp, compiled for the host too: tests/tools/dt_core_test.cpp).
//             Reference arithmetic throughout: intersection in fp64, narrowed to
//             fp32 (`T s = f(...)`, :161), `s <= z[k]` pops (:162), read-out
//             `z[k+1] < os` with the int promoted to T (:174) and the value
//             evaluated in fp64 (:175).  Lines are read coalesced (line-contiguous
//             input) and the result is written TRANSPOSED (element q of line i at
//             q*nlines+i), so the x pass (rows) feeds the y pass (columns)
//             line-contiguously and the y pass lands in row-major layout again.
// k_reduce    Math::reduceMax over the child mixtures for every parent mixture (Ik;
//             Ix / Iy are composed from the DT's own pointer planes at back-tracking
//             time, DistanceTransform.hpp:233-244) and the in-order accumulation
//             into the parent score (DynamicProgram.cpp:134-156).
// k_root      root bias + reduceMax (:163-171), strict threshold (:208) and
//             compaction of the hits.
// k_backtrack argmin (:219-245): one block per candidate, one lane per part, depth by depth.
#include <algorithm>
#include <type_traits>
#include "pbd_internal.hpp"
#include "dt_core.hpp"

// pointers that come out of descriptors in memory are generic to the compiler (flat loads and stores, which also
// count as LDS traffic for s_waitcnt): say that they are global
#define GP(U) const __attribute__((address_space(1))) U*
#define GPW(U) __attribute__((address_space(1))) U*

// debug: per-phase timestamps (100 MHz wall clock) of block 0 of the last k_dt_pass launch
#ifdef PBD_PROBES
__device__ unsigned long long pbd_dt_dbg[8];
#define DT_COUNT_REDO() atomicAdd(&pbd_dt_dbg[7], 1ull)
// block trace: (start, end) wall clock and hardware id of every block of the first 40 launches since the last read
#define DT_TRACE_L 40
#define DT_TRACE_B 4096
__device__ unsigned long long pbd_dt_trace[DT_TRACE_L][DT_TRACE_B][8];   // [0..6]: the DT_STAMP phases, [7]: end
__device__ unsigned pbd_dt_trace_hw[DT_TRACE_L][DT_TRACE_B];
__device__ int pbd_dt_trace_launch = DT_TRACE_L;   // off until PBD_DT_TRACE is set
static int g_dt_trace_seq = 0;
#define DT_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); if (blockIdx.x == 0) pbd_dt_dbg[i] = now_; \
    if (i && pbd_dt_trace_launch < 40 && blockIdx.x < 4096) pbd_dt_trace[pbd_dt_trace_launch][blockIdx.x][i] = now_; } } while (0)
#define DT_TRACE(k) do { if (threadIdx.x == 0 && pbd_dt_trace_launch < DT_TRACE_L && blockIdx.x < DT_TRACE_B) { \
    pbd_dt_trace[pbd_dt_trace_launch][blockIdx.x][k ? 7 : 0] = wall_clock64(); \
    if (k == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); pbd_dt_trace_hw[pbd_dt_trace_launch][blockIdx.x] = (hw & 0xffffff) | (xcc << 24); } } } while (0)
int dt_debug_trace(unsigned long long* t, unsigned* hw, int* nlaunch) {
  hipMemcpyFromSymbol(t, HIP_SYMBOL(pbd_dt_trace), sizeof(unsigned long long) * DT_TRACE_L * DT_TRACE_B * 8);
  hipMemcpyFromSymbol(hw, HIP_SYMBOL(pbd_dt_trace_hw), sizeof(unsigned) * DT_TRACE_L * DT_TRACE_B);
  *nlaunch = g_dt_trace_seq;
  g_dt_trace_seq = 0;
  return 0;
}
void dt_debug_read(unsigned long long* out) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pbd_dt_dbg), sizeof(unsigned long long) * 8);
  const unsigned long long z = 0;
  hipMemcpyToSymbol(HIP_SYMBOL(pbd_dt_dbg), &z, sizeof(z), 7 * sizeof(unsigned long long));   // [7]: lines redone sequentially since the last read
}
#else
#define DT_STAMP(i) do { } while (0)
#define DT_COUNT_REDO() do { } while (0)
#define DT_TRACE(k) do { } while (0)
void dt_debug_read(unsigned long long* out) { for (int i = 0; i < 8; ++i) out[i] = 0; }
#endif

// LDS per block: a header (per-line and per-lane descriptors, segment table), ONE table of exact reciprocals
// 1/dx, dx < len (double[S], shared by all lines whatever their map: dt_core.hpp), and per line
// {(y, z) : T2[S]; B : u8[S] (S <= 256) or u16[S]}.
// 9 bytes per line element for float: the lines resident on a CU are bounded by these bytes.
#define DT_SEGS 72                                   // SEG entries: P + 1 <= 65 starts, then {0, len} for a line redone as one segment
__host__ __device__ inline size_t dt_hdr_bytes(int nt, int ts, int its, int lpb) {   // per line 16 B, per lane T + 4 IT
  return ((size_t)lpb * 16 + DT_SEGS * 4 + (size_t)nt * (ts + 4 * its) + 15) & ~(size_t)15;
}
size_t dt_lds_bytes(int stride, int lpb, int ts, int nt) {   // ts = sizeof(T): (y, z) is a float or a double pair
  const int its = stride <= 256 ? 1 : 2;
  return (size_t)lpb * stride * (2 * ts + its) + dt_hdr_bytes(nt, ts, its, lpb) + (((size_t)stride + 1) & ~(size_t)1) * 8 + 16;
}

// ---- message fold (fold mode) -------------------------------------------------------------------------------
// The reference sends a child's message as soon as the child has been transformed (src/DynamicProgram.cpp:134-156):
//   for every parent mixture m: weighted[k] = sdt_k + bias(k)[m] (:139), (maxv, maxi) = reduceMax (:143: init -inf,
//   strict >, first maximum wins; K == 1: copy), Ik = maxi (:150), parent.score(m) += maxv (:156, the parent's score
//   is a copy of its raw response until its first message, :155) — children in DESCENDING index order (:95).
// Nothing reads the parent's accumulated score before the parent's own transform (or, for a root, the root
// reduction), so the fold is done by the consumer: acc[m] enters with the parent's raw response at the cell and
// leaves as ((raw + msg_c1) + msg_c2) ... — the same float operations in the same order — and the accumulated
// planes are never stored.  The K child values of a cell are loaded once for all L parent mixtures.
// M = upper bound of the mixture counts involved (register arrays; a launch is instantiated for the bound of its model).
// U cells per lane at once: all child loads of the U cells are issued before the first use, and the K x L bias
// block of the child (wave-uniform: scalar loads) is fetched in one straight-line batch — a load inside a (uniform)
// branch costs one full scalar-memory round trip per branch, which is what made a first version's loader 5x slower
// than the plain one.  Everything below is branch-free except the loop over the children.
// A child's plane pointers wait in a VECTOR register, one quad-word per lane (lane k < 8: sdt[k], lanes >= 8: ok), and come out by v_readlane where they are
// used: fetched by scalar loads at their use (rounds 3-5) every child began with a memory round trip of its own — descriptor -> pointers -> planes —, and
// fetched early by scalar loads they do not fit the scalar register file beside the bias block (hipcc then serialises the loads, one wait each: tried).  The
// caller fetches child 0's with whatever else it reads from descriptors; child c + 1's are fetched behind child c's vector loads.  Every lane of the
// wavefront must be active where fold_child_qw runs.
static_assert(offsetof(FoldChild, ok) == 8 * PBD_FOLD_MAXMIX && PBD_FOLD_MAXMIX == 8, "fold_child_qw: sdt[8] then ok");
__device__ __forceinline__ unsigned long long fold_child_qw(const FoldChild* C) {
  return ((GP(unsigned long long))C)[min((int)(threadIdx.x & 63u), 8)];
}
__device__ __forceinline__ GP(char) fold_qw_lane(unsigned long long v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return (GP(char))(((unsigned long long)hi << 32) | lo);
}
template <typename T, int M, int U>
__device__ __forceinline__ void fold_children(const FoldJob* __restrict__ J, const float* __restrict__ /*biasw*/, const unsigned (&off)[U],
                                              unsigned HW, int L, T (&acc)[U][M], unsigned long long cv, int nch) {   // nch = J->nch (k_dt_pass has it from the task's extension record)
  unsigned ob[U];                                        // byte offsets of the cells inside a plane of T (< 2^32, plan_frame)
#pragma unroll
  for (int u = 0; u < U; ++u) ob[u] = off[u] * (unsigned)sizeof(T);
  int c = 0;
  do {                                                   // (a fold job has at least one child: a loop that may run zero times made the
                                                         // compiler wait, after it, for the children's Ik STORES before the LDS stores)
    const FoldChild& C = J->ch[c];
    // everything the child contributes is fetched up front, in straight-line code: the K planes' values of the U
    // cells (uniform base + 32-bit byte offset: no vector arithmetic per load) and the dense K x L bias block
    GPW(uint8_t) okp = (GPW(uint8_t))fold_qw_lane(cv, 8);
    T sd[U][M];
    // (the offsets pass through an empty asm: hoisted out of the loop over the children they would be kept zero-extended to
    // 64 bits, and every load / store would pay a 64-bit vector add instead of using the scalar-base + 32-bit-offset form)
    unsigned obc[U], offc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      obc[u] = ob[u]; offc[u] = off[u];
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(obc[u]), "+v"(offc[u]));
#endif
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
      GP(char) pl = fold_qw_lane(cv, k);                 // entries beyond K repeat plane K - 1 (plan): never predicated
#pragma unroll
      for (int u = 0; u < U; ++u) sd[u][k] = *(GP(T))(pl + obc[u]);
    }
    cv = fold_child_qw(&J->ch[min(c + 1, nch - 1)]);      // the next child's pointers (after the last child: its own again, unused)
    float bias[M][M];
    // (wave-uniform: scalar loads.  Fetching the block with vector loads instead — it overflows the scalar register file and
    // part of it is spilled to vector-register lanes — was measured 9 % slower per fold launch: twelve more vector-memory
    // instructions per child in front of the block's dependent chain)
#pragma unroll
    for (int k = 0; k < M; ++k)
#pragma unroll
      for (int m = 0; m < M; ++m) bias[k][m] = C.bias[k][m];
    unsigned okoff[M];                                   // offset of plane m of the child's Ik planes (uniform).  Columns beyond L repeat column
#pragma unroll                                           // L - 1 (plan) and land on plane L - 1 again: the same byte stored twice, no predicate
    for (int m = 0; m < M; ++m) okoff[m] = (unsigned)min(m, L - 1) * HW;
    // Math::reduceMax's K == 1 shortcut copies (Math.hpp:154-158): maxv = the one weighted map — NaN and -inf included —,
    // maxi = 0.  A wave-uniform SELECT at the end, not a branch: a branch here made the compiler wait for plane 0's loads
    // (the code both sides share) before it issued the loads of the other planes — two memory round trips per child.
    const bool copy1 = C.K == 1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      T v[M];
      int bi[M];
      T w0[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        // k = 0 first: Math::reduceMax starts from -inf and takes strict > (first maximum wins): a NaN score leaves -inf
        w0[m] = sd[u][0] + bias[0][m];                    // DynamicProgram.cpp:139
        v[m] = w0[m] > (T)-INFINITY ? w0[m] : (T)-INFINITY;
        bi[m] = 0;
      }
#pragma unroll
      for (int k = 1; k < M; ++k) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          // (no `k < K` test: planes / bias rows beyond K repeat mixture K - 1, whose value cannot be strictly greater than the
          // maximum it has already been folded into)
          const T wv = sd[u][k] + bias[k][m];
          const bool take = wv > v[m];                  // strict >: first max wins
          bi[m] = take ? k : bi[m];
          v[m] = take ? wv : v[m];
        }
      }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        // Ik (:150).  Unpredicated: a lane past the block's last cell works on that cell again (its offset was clamped) and
        // stores the same byte once more
        *(okp + (okoff[m] + offc[u])) = (uint8_t)bi[m];                // (cells * planes < 2^32, plan_frame; K == 1: rows beyond 0 repeat row 0, bi stays 0)
        acc[u][m] = acc[u][m] + (copy1 ? w0[m] : v[m]);                // parent.score += maxv (:156), child order kept
      }
    }
  } while (++c < nch);
}

// One block = NT lanes (one, two or four wavefronts) = up to g.lpb lines of one group (lpb chosen per group so that every
// block of the launch fits the same LDS budget: long lines -> fewer lines per block -> many more blocks).  LDS
// capacity leaves most lanes without a line of their own, so the NT / lpb lanes that share a line each scan one
// SEGMENT of it concurrently and the segments are stitched into the sequential result (dt_core.hpp):
// lane = p * lpb + line.
// FOLD: the block's lines are nrows consecutive rows x the K mixtures of one part (line = mixture * nrows + row, so
// that neighbouring lanes write neighbouring rows of one transposed plane) and the loader builds them from the part's
// raw responses and its children's messages (fold_children).
template <typename T, typename IT, int FM>   // FM: 0 = plain lines, else fold with at most FM mixtures per part
__device__ __forceinline__ void dt_block(char* smem, const DtTask& t, const DtGroup& g, const DtMap* __restrict__ maps,
                                         const FoldJob* __restrict__ folds, const float* __restrict__ biasw,
                                         unsigned long long rawq, unsigned long long c0qw) {   // FOLD: the task's extension record, one quad-word per lane (k_dt_pass)

  constexpr bool FOLD = FM > 0;
  const int lane = threadIdx.x, NT = blockDim.x;
  const int len = g.len, S = g.stride, lpb = g.lpb;
  typedef DtPair<T> P2;
  constexpr bool EX = sizeof(T) == 8;          // DistanceTransform<double>: s is not narrowed, every intersection takes the IEEE division
  // float maps whose weights are converted floats on lines of at most DT_FUSE_MAXLEN elements (every map of a detector; the planner says so per
  // group): the numerator's products are exact, so they fuse into their additions (dt_core.hpp: dt_isect, FUSED) — block-uniform
  const bool fz = !EX && (g.fused & DT_G_FUSED) != 0;
  const T** lptr = (const T**)smem;            // [lpb] source pointer of each line of this block (plain)
  int* FLAG = (int*)(smem + lpb * 8);          // [lpb] per line: redo sequentially (suspect quotient / lost stitch invariant)
  int* FIX = FLAG + lpb;                       // [lpb] per line: the lowest stale boundary of the line in an even validation round
  int* FIX2 = (int*)smem;                      // [lpb] ... in an odd round (the line-pointer table is dead after the loader)
  int* SEG = FIX + lpb;                        // [P + 1 <= 65] start of every segment (len and P are uniform over the block), [DT_SEGS - 2..]: {0, len}
  int* ANY = SEG + DT_SEGS - 6;                // [2] a line has a stale stitch (even / odd validation round)
  int* NFLAG = SEG + DT_SEGS - 4;              // [1] a line of the block is flagged for the sequential redo
  T* ZLO = (T*)(SEG + DT_SEGS);                // [NT] per lane (p * lpb + line): z of the segment's lowest surviving element — what the read-out needs; while
                                               // validation rounds run (rare path): that element's LOCAL z, before the stitch patched it (the stitch's lane keeps
                                               // it in a register and stores it only if the block has a stale stitch: 4 B per lane less LDS than a table of its
                                               // own — at 640x480 the person model's fold launches of a single frame drop from 1 040 to 1 016 blocks, under the
                                               // 1 024 the chip holds at once)
  IT* FT = (IT*)(ZLO + NT);                    // [NT] per lane: that element
  IT* BELOW = FT + NT;                         // [NT] the element FT sits on
  IT* DMIN = BELOW + NT;                         // [NT] lowest element the segment's speculative stitch tested
  IT* BSAVE = DMIN + NT;                       // [NT] local link of FT (before the patch)
  double* RDX = (double*)(smem + dt_hdr_bytes(NT, sizeof(T), sizeof(IT), lpb));   // [S] 1/dx
  const int nl = t.nl;                         // lines of this block
  const int nrows = FOLD ? nl / g.nmaps : 0;   // FOLD: rows of this block
  P2* YZ = (P2*)(RDX + ((S + 1) & ~1));        // [lpb][S] (16-byte aligned) .x: line values (never modified); .y: z of the element when pushed
  IT* B = (IT*)(YZ + lpb * S);                 // [lpb][S] element below on the stack when pushed; later: element above (read-out)
  // line i of a plain block = line (t.l0 + i) mod nlines of map t.m0 + (t.l0 + i) / nlines; the quotients by wave-uniform numbers
  // come as multiply-high constants from the plan (DtGroup)
  auto map_line = [&](int i, int& mi, int& li) {
    const int x = t.l0 + i;
    const int q = g.nlines > 1 ? (int)__umulhi((unsigned)x, g.magic_nlines) : x;
    mi = t.m0 + q; li = x - __mul24(q, g.nlines);
  };
  // plain: a block whose lines lie back to back in memory (DtTask::src0: the y pass always, plan_frame) loads them as ONE run — no line
  // pointers, hence no map descriptor in front of the loads (a dependent memory round trip of ~1 us) and no barrier in front of the loader
  const bool contig = !FOLD && t.src0 != nullptr;   // (block-uniform)
  if (lane < nl) {
    if (!FOLD && !contig) {
      int mi, li;
      map_line(lane, mi, li);
      const DtMap& mp0 = maps[g.map0 + mi];
      lptr[lane] = (const T*)mp0.src + (size_t)li * len;
    }
    FLAG[lane] = 0;
  }
  if (lane < lpb) FIX[lane] = 0x7fffffff;
  const int nsub = g.nsub;                       // lanes per line
  const int P = g.P;                             // segments per line (dt_segments(nsub, len))
  if (lane <= P) SEG[lane] = P > 1 ? (int)__umulhi((unsigned)__mul24(lane, len), g.magic_P) : lane * len;   // dt_seg_start(lane, P, len)
  if (lane == 0) { SEG[DT_SEGS - 2] = 0; SEG[DT_SEGS - 1] = len; ANY[0] = 0; *NFLAG = 0; }
  const int p = lpb > 1 ? (int)__umulhi((unsigned)lane, g.magic_lpb) : lane, line = lane - __mul24(p, lpb);
  const bool mine = line < nl && p < nsub;
  // line -> (map, line of the map): plain: map-major; FOLD: mixture-major inside the block's rows
  int mi = 0, li = 0;
  if (mine) {
    if (FOLD) {
      const unsigned mrows = nrows > 1 ? (0xFFFFFFFFu / (unsigned)nrows + 1u) : 0u;   // (wave-uniform: scalar arithmetic)
      mi = nrows > 1 ? (int)__umulhi((unsigned)line, mrows) : line;
      li = t.g0 + (line - __mul24(mi, nrows));
    } else {
      map_line(line, mi, li);
    }
  }
  // the lane's map descriptor (weights, destination, pointer plane): fetched HERE, in front of the loader, so that its memory
  // round trip overlaps the loader's instead of standing between the loader's barrier and the scan
  DtMap mp;
  P2* YZl = YZ + line * S;
  IT* Bl = B + line * S;
  mp = maps[g.map0 + mi];   // (every lane — lanes without a line read map 0 of the group —: behind an `if (mine)` the wait for the fold's extension record, at the
                            //  join of the two paths, also waited for these loads — a memory round trip in front of the loader)
  if (!FOLD && !contig) __syncthreads();
  DT_STAMP(1);
  if constexpr (FOLD) {
    // ---- fold loader: U (row, element) cells per lane and step, all mixtures of the part at once; the loads of the U
    // cells are issued back to back (one memory round trip per step, like the plain loader's batches) ----
    constexpr int M = FM > 0 ? FM : 1;
    constexpr int U = sizeof(T) == 8 ? 1 : 3;
    const FoldJob* J = folds + g.fold;
    const int L = g.nmaps;
    const unsigned HW = (unsigned)g.nlines * (unsigned)len;     // cells of the level (< 2^31, plan_frame)
    // the part's raw response planes (entries beyond L repeat plane L - 1) and the first child's plane pointers (c0qw: fold_children) come from the task's
    // extension record, fetched by k_dt_pass BESIDE the task descriptor: the loader's first loads are one memory round trip behind the kernel's entry, not
    // three (descriptor -> map table / fold job -> planes; round 6)
    GP(char) srcp[M];
#pragma unroll
    for (int m = 0; m < M; ++m) srcp[m] = fold_qw_lane(rawq, m);
    const int nch0 = __builtin_amdgcn_readlane((int)(unsigned)c0qw, 9);   // FoldJob::nch
    const int n = nrows * len;
    // the block's cells are rows t.g0 .. t.g0 + nrows - 1 of the level: CONTIGUOUS in every plane, cell ec of the block at
    // plane offset t.g0 * len + ec (no division for the loads); its (row, column) — the LDS slot — by multiply-high
    const unsigned cell0 = (unsigned)t.g0 * (unsigned)len;
    const unsigned magic = len > 1 ? (0xFFFFFFFFu / (unsigned)len + 1u) : 0u;   // ec / len = umulhi(ec, magic), exact for ec * len < 2^32 (len = 1: ec itself)
    for (int e0 = 0; e0 < n; e0 += NT * U) {
      T acc[U][M];
      unsigned off[U];
      int slot[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ec = min(e0 + u * NT + lane, n - 1);
        off[u] = cell0 + (unsigned)ec;
        const unsigned obu = off[u] * (unsigned)sizeof(T);             // byte offset inside a plane (< 2^32, plan_frame): uniform base + 32-bit offset
#pragma unroll
        for (int m = 0; m < M; ++m) acc[u][m] = *(GP(T))(srcp[m] + obu);
        const int jj = len > 1 ? (int)__umulhi((unsigned)ec, magic) : ec;
        slot[u] = __mul24(jj, S) + (ec - __mul24(jj, len));        // LDS element of mixture 0's line of that row
      }
      // reciprocal table 1/dx: one IEEE division per entry, spread over the lanes — while the loads are in flight (round 6: copying the
        // entries from a per-device table instead — 30 fewer vector instructions per lane — measured 1.7 % SLOWER in batches: the divisions
        // hide under the loads, a dependent global load in front of the LDS write does not)
      if constexpr (!EX) {
        if (e0 == 0)
          for (int dx = lane; dx < len; dx += NT) RDX[dx] = 1.0 / (double)dx;   // entry 0 is never read
      }
      fold_children<T, M, U>(J, biasw, off, HW, L, acc, c0qw, nch0);
      const int mstride = nrows * S;                                 // LDS elements between the lines of consecutive mixtures of a row
      // (unpredicated like the plain loader: a lane past the last cell holds the last cell's values and stores them once more.  Round 6
      //  measured the alternative — wavefronts whose slot holds clamped repeats only skip the mixture reduce —: 0.7 % faster alone, 1.6 %
      //  slower in batches, profiles/r06/r06_session16_*)
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int m = 0; m < M; ++m)      // (columns beyond L repeat mixture L - 1: the same value stored to its slot once more, no branch)
          YZ[min(m, L - 1) * mstride + slot[u]].x = acc[u][m];
      }
    }
  } else {
    // coalesced load of the nl lines: lane-linear over the block's elements, line after line (consecutive lines of a
    // map are contiguous in memory, so the loads stay coalesced across line ends and short lines — 100+ lines of
    // 6-20 elements on the small levels — fill the lanes like long ones).  Batches of LB independent loads are
    // issued before the first wait (addresses are clamped instead of predicated: a predicated load makes hipcc
    // branch and wait per element, which serialises one full memory round trip per 256 B).  The kernel is bound by
    // vector-instruction issue, so an element's (line, position) is worked out ONCE — its LDS address waits in a register
    // for the value (the kernel's occupancy is set by LDS, not by registers) — and the division is a multiply-high.
    const int n = nl * len;
    constexpr int LB = sizeof(T) == 8 ? 12 : 24;   // loads in flight per lane (a 25 KB / 128-lane block of float lines: <= 22 elements per lane; 40 KB / 256 lanes: <= 18)
    if (len > 1) {
      // f / len = umulhi(f, magic), magic = ceil(2^32 / len): exact for f * len < 2^32
      const unsigned magic = 0xFFFFFFFFu / (unsigned)len + 1u;
      auto load_lines = [&](auto run) {             // run: the block's elements are one contiguous run at t.src0
        constexpr bool RUN = decltype(run)::value;
        GP(T) s0 = (GP(T))t.src0;
        for (int f0 = 0; f0 < n; f0 += LB * NT) {
          T r[LB];
          int la[LB];
#pragma unroll
          for (int j = 0; j < LB; ++j) {
            const int f = min(f0 + j * NT + lane, n - 1);
            const int i = (int)__umulhi((unsigned)f, magic);
            const unsigned pos = (unsigned)(f - __mul24(i, len));
            if constexpr (RUN) r[j] = s0[f];
            else r[j] = ((GP(T))lptr[i])[pos];
            la[j] = __mul24(i, S) + (int)pos;
          }
          // reciprocal table 1/dx: one IEEE division per entry, spread over the lanes — while the loads are in flight (round 6: copying the
        // entries from a per-device table instead — 30 fewer vector instructions per lane — measured 1.7 % SLOWER in batches: the divisions
        // hide under the loads, a dependent global load in front of the LDS write does not)
          if constexpr (!EX) {
            if (f0 == 0)
              for (int dx = lane; dx < len; dx += NT) RDX[dx] = 1.0 / (double)dx;   // entry 0 is never read
          }
          // (unpredicated: a lane past the block's last element holds that element's value again — its load address was
          // clamped — and stores it once more; a predicate per element made hipcc emit 24 nested exec-mask regions)
#pragma unroll
          for (int j = 0; j < LB; ++j) YZ[la[j]].x = r[j];
        }
      };
      if (contig) load_lines(std::true_type{});
      else load_lines(std::false_type{});
    } else {   // lines of ONE element (a 1-wide level, or pbd_dt2d on a vector): element f is line f
      for (int f = lane; f < n; f += NT) YZ[f * S].x = ((GP(T))lptr[f])[0];
      if constexpr (!EX) { if (lane == 0) RDX[0] = 0.0; }
    }
  }
  __syncthreads();
  DT_STAMP(2);

  // ---- local scans: the envelope of every segment (DistanceTransform.hpp:156-170 on the segment alone) ----
  if (mine && p < P) {
    const bool sus = fz ? dt_seg_scan<EX, !EX, T, IT>(YZl, Bl, RDX, mp.r2a, SEG[p], SEG[p + 1], mp.a, mp.b)
                        : dt_seg_scan<EX, false, T, IT>(YZl, Bl, RDX, mp.r2a, SEG[p], SEG[p + 1], mp.a, mp.b);
    if (sus) { FLAG[line] = 1; *NFLAG = 1; }
  }
  __syncthreads();
  DT_STAMP(3);
  // ---- stitch the segments into the sequential result: every boundary by its own lane, concurrently ----
  T zs_mine = (T)0;                                // the local z of this lane's F (a redo restores it: dt_stitch_redo)
  const bool stitched = mine && p >= 1 && p < P && !FLAG[line];
  if (stitched) {
    int f, dmin, bs;
    T zs;
    const bool bad = fz ? dt_stitch1<EX, !EX, T, IT>(YZl, Bl, RDX, mp.r2a, SEG[p], SEG[p + 1], mp.a, mp.b, f, dmin, zs, bs)
                        : dt_stitch1<EX, false, T, IT>(YZl, Bl, RDX, mp.r2a, SEG[p], SEG[p + 1], mp.a, mp.b, f, dmin, zs, bs);
    FT[lane] = (IT)f; DMIN[lane] = (IT)dmin; zs_mine = zs; BSAVE[lane] = (IT)bs;
    if (bad) { FLAG[line] = 1; *NFLAG = 1; }
  }
  if (mine && p == 0) FT[lane] = (IT)0;
  __syncthreads();
  // ---- validate the speculation (dt_core.hpp): a stitch is final iff everything it tested below its segment lies strictly
  // above the lowest element its left neighbour's stitch left; every lane judges its own and publishes what the read-out needs
  // to know about its segment.  Usually no lane is stale and that is all.  Else: rounds — the LOWEST stale boundary of a line
  // (everything to its left is final) is redone, the others are judged again against their neighbour's new F.  The redo is
  // done by the line's FIRST lane (p = 0: the block's first wavefront or two): the stale boundaries of different lines sit in
  // different wavefronts, and every wavefront that holds one would run the whole stitch loop for one live lane.
  // FIX / FIX2 [line] = lowest stale boundary of the line in an even / odd round, ANY[parity] = some line has one.
  {
    const bool cand = mine && p >= 2 && p < P;
    const bool live = mine && p < P && !FLAG[line];
    const int fspec_prev = cand ? (int)FT[lane - lpb] : 0;       // F_spec[p - 1]: what the neighbour's SPECULATIVE stitch patched
    if (cand && live && dt_stitch_stale((int)DMIN[lane], fspec_prev, fspec_prev)) { atomicMin(&FIX[line], p); ANY[0] = 1; }
    if (live) {
      const int f = (int)FT[lane];
      BELOW[lane] = Bl[f];
      ZLO[lane] = YZl[f].y;
    }
    __syncthreads();
    DT_STAMP(6);
    if (ANY[0]) {                                  // (block-uniform; rare)
      if (stitched) ZLO[lane] = zs_mine;           // the table holds the stitches' local z while the rounds run (the segments' entries are published again below)
      __syncthreads();
      for (int round = 0;; ++round) {
        int* FIXr = (round & 1) ? FIX2 : FIX;
        int* FIXn = (round & 1) ? FIX : FIX2;
        if (mine && p == 0 && FIXr[line] != 0x7fffffff) {
          const int ps = FIXr[line], ls = __mul24(ps, lpb) + line;
          int f, bs = (int)BSAVE[ls];
          T zs = ZLO[ls];
          const bool bad = fz ? dt_stitch_redo<EX, !EX, T, IT>(YZl, Bl, RDX, mp.r2a, SEG[ps], SEG[ps + 1], mp.a, mp.b, (int)FT[ls], f, zs, bs)
                              : dt_stitch_redo<EX, false, T, IT>(YZl, Bl, RDX, mp.r2a, SEG[ps], SEG[ps + 1], mp.a, mp.b, (int)FT[ls], f, zs, bs);
          FT[ls] = (IT)f; BSAVE[ls] = (IT)bs;
          DMIN[ls] = (IT)SEG[ps];                  // final: never stale again (every F to its left lies below its segment) — its saved z is not needed again
          if (bad) { FLAG[line] = 1; *NFLAG = 1; }
        }
        if (lane < lpb) FIXn[lane] = 0x7fffffff;   // the next round's tables (last read a round ago)
        if (lane == 0) ANY[(round + 1) & 1] = 0;
        __syncthreads();
        if (cand && !FLAG[line] && dt_stitch_stale((int)DMIN[lane], fspec_prev, (int)FT[lane - lpb])) { atomicMin(&FIXn[line], p); ANY[(round + 1) & 1] = 1; }
        __syncthreads();
        if (!ANY[(round + 1) & 1]) break;
      }
      if (mine && p < P && !FLAG[line]) {          // the segments' entries once more (a redo moves F)
        const int f = (int)FT[lane];
        BELOW[lane] = Bl[f];
        ZLO[lane] = YZl[f].y;
      }
      __syncthreads();
    }
  }
  // the rare line with a suspect quotient or a lost invariant is redone as a whole by one lane (IEEE divisions) and read out
  // as one segment (NFLAG: lines flagged in this block)
  if (*NFLAG) {                                    // (block-uniform)
    if (mine && p == 0 && FLAG[line]) {
      DT_COUNT_REDO();
      dt_seg_scan<true, false, T, IT>(YZl, Bl, RDX, mp.r2a, 0, len, mp.a, mp.b);
      BELOW[line] = Bl[0];
      ZLO[line] = YZl[0].y;
    }
    __syncthreads();
  }
  DT_STAMP(4);

  // ---- read out (:172-178) ----
  // Output q of a line depends only on the finished stack, so the 64/lpb lanes of a line each take a sub-range of
  // the outputs, all sub-ranges stepping q in lockstep — downwards, along the "below" links: scores go out
  // transposed (lanes of one sub-range = consecutive lines -> coalesced); pointers go straight to their plane:
  // transposed like the scores in the y pass, natural (2-byte runs per lane, merged in L2) in the x pass — no
  // LDS staging, that space holds more lines.
  if (mine) {
    const double a = mp.a, b = mp.b;
    // pointers: transposed like dst (y pass -> natural layout) or natural (x pass)
#ifdef PBD_NAT_FROM_MAP   // (A/B builds: the form before round 6's session 31)
    const bool nat = mp.ptr_natural != 0;
#else
    const bool nat = (g.fused & DT_G_NATURAL) != 0;   // (= mp.ptr_natural, uniform over a group: DtGroup::fused)
#endif
    int16_t* pp = mp.ptr + (nat ? (size_t)li * len : (size_t)li);
    const int pst = nat ? 1 : g.nlines;
    const int chunk = g.chunk;                   // ceil(len / nsub)
    const int q0 = p * chunk, q1 = min(len, q0 + chunk);
    if (q0 < q1) {
      int os = mp.os + q1 - 1;
      const bool whole = FLAG[line] != 0;
      int e = dt_cover<T, IT>(YZl, Bl, whole ? SEG + DT_SEGS - 2 : SEG, whole ? 1 : P, BELOW + line, ZLO + line, lpb, os);
      P2 eyz = YZl[e];
      // the link below the current piece waits in a register, so a step down issues its two LDS reads (the piece and ITS
      // link) at once: one round trip per step.  (Measured and rejected in round 4:
      // the read-out as a flat state machine like the scans, one output OR one step per iteration — 0.371 ms per frame
      // instead of 0.329: most outputs need no step, and the flat form makes every one of them wait for a speculative read.)
      // The bottom of the stack (z = -inf, linked to itself) ends every walk.
      int nx = (int)Bl[e];
      // Round 6, session 39: the piece below and ITS link wait in registers again — a step down is register moves, and the LDS reads it issues (the piece after
      // that) are only needed one step later: in a single frame's launches the read-out is a chain of such steps (almost every output has a lane of the wavefront
      // stepping), dp_min alone 0.494 -> 0.489 ms, 0.296 -> 0.295 in batches (round 4 had removed the prefetch for four vector instructions per step, when issue
      // slots were all that counted).
      P2 nyz = YZl[nx];
      int nnx = (int)Bl[nx];
      const int nlines = g.nlines;
      GPW(T) dp = (GPW(T))mp.dst + li + (size_t)(q1 - 1) * nlines;      // running output pointers: no 64-bit multiply per element
      GPW(int16_t) ppq = (GPW(int16_t))pp + (size_t)(q1 - 1) * pst;
      const int os_end = mp.os + q0;           // the sub-range's first output (the loop counts the shifted position down to it)
      ++os;
      auto outputs = [&](auto fused) {         // fused: a * d^2 and b * d are exact (|d| < 2^14, float-born weights): their sum is one fma
        constexpr bool FZ = decltype(fused)::value;
        do {                                   // (q0 < q1: at least one output)
          --os;
          const T fos = (T)os;                 // `z[k+1] < os`: int promoted to T (:174)
          while (!(eyz.y < fos)) { e = nx; eyz = nyz; nx = nnx; nyz = YZl[nx]; nnx = (int)Bl[nx]; }
          const double d = (double)(os - e);   // |d| < 2^15: d * d is exact in fp64 (the reference squares the int)
          const double ad2 = a * (d * d);
          *dp = (T)((FZ ? fma(b, d, ad2) : (ad2 + b * d)) + (double)eyz.x);
          *ppq = (int16_t)e;
          dp -= nlines; ppq -= pst;
        } while (os > os_end);
      };
      if constexpr (!EX) { if (fz) outputs(std::true_type{}); else outputs(std::false_type{}); }
      else outputs(std::false_type{});
    }
  }
  DT_STAMP(5);
}

// (round 5, measured and rejected: the register allocation forced to 96 — launch bound 5 — so that two DT wavefronts fit on a SIMD beside a
//  304-register wavefront of the split-product bank: dp_min 0.339-0.344 ms per frame against 0.330-0.337, 2 029-2 056 against 2 089-2 118 frames/s;
//  the fold loader fetching the next child's planes under this child's arithmetic: 0.333-0.344 against 0.337-0.342, 0.585-0.588 against 0.581 alone)
template <typename T, int FM>
__global__ __launch_bounds__(256, 3) void k_dt_pass(const DtTask* __restrict__ tasks, const DtMap* __restrict__ maps,
                                                    const FoldJob* __restrict__ folds, const float* __restrict__ biasw,
                                                    const unsigned long long* __restrict__ foldx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Issue priority 1 for the whole block (round 6): where a distance-transform wavefront shares a SIMD with wavefronts of another batch's filter
  // bank (priority 0: they wait for the matrix pipe most of the time) or HOG kernel, its vector instructions go first — the DP chain is the longest
  // dependent chain of a frame.  Interleaved runs on two boxes: +1.2 % frames/s with single-frame calls in flight (1 921 -> 1 944, 1 919 -> 1 942), +0.6 ... 1.1 % in
  // batches; priorities 1 / 2 / 3 measure the same, the filter bank's K loop at 1 / 3 instead: no change (profiles/r06/r06_session28_*, _session29_*).
  // Alone on the chip the DT is unaffected (rounds 4's s_setprio experiments were phase-wise INSIDE the block, against its own co-resident blocks: slower).
#if PBD_DT_PRIO > 0
  __builtin_amdgcn_s_setprio(PBD_DT_PRIO);
#endif
  DT_STAMP(0);
  DT_TRACE(0);
#if PBD_DT_TASK_PREFETCH > 0 && defined(__HIP_DEVICE_COMPILE__)
  // Touch the descriptor of the block PBD_DT_TASK_PREFETCH tasks ahead (same XCD for a multiple of 8) so that ITS scalar fetch — the first of a block's
  // dependent memory round trips (descriptor -> lines) — finds the line in this XCD's L2.  A VECTOR load (an outstanding scalar load would be waited for
  // together with the block's own descriptor); the zero offset passes through an empty, non-volatile asm: a volatile one makes hipcc fetch every descriptor
  // with vector loads (it counts as a clobber of memory).  Round 6, sessions 30 / 31: 64 ... 256 ahead +0.1 ... 1 % frames/s, -0.4 % for a frame alone; 1 024 ahead: nothing
  // (the line does not survive in the L2 that long)
  unsigned pf_off = 0;
  asm("" : "+v"(pf_off));
  const unsigned pf_val = *(const unsigned*)((const char*)(tasks + min(blockIdx.x + (unsigned)PBD_DT_TASK_PREFETCH, gridDim.x - 1u)) + pf_off);
#endif
  // fold launches: the task's extension record (PBD_FOLDX_QW quad-words: raw plane pointers [8], first child's plane pointers [8], its Ik base, the number of children), lane l holding
  // quad-word l of each half — vector loads whose address depends on blockIdx only, in flight together with the scalar fetch of the task
  unsigned long long rawq = 0, c0qw = 0;
  if constexpr (FM > 0) {
    GP(unsigned long long) fx = (GP(unsigned long long))foldx + (size_t)blockIdx.x * PBD_FOLDX_QW;
    const int l = (int)(threadIdx.x & 63u);
    rawq = fx[min(l, 7)];
    c0qw = fx[8 + min(l, 9)];
  }
  const DtTask t = tasks[blockIdx.x];
  const DtGroup& g = t.g;
  if (g.stride <= 256) dt_block<T, unsigned char, FM>(smem, t, g, maps, folds, biasw, rawq, c0qw);    // stack indices < 255 fit a byte
  else dt_block<T, unsigned short, FM>(smem, t, g, maps, folds, biasw, rawq, c0qw);
  DT_TRACE(1);
#if PBD_DT_TASK_PREFETCH > 0 && defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" :: "v"(pf_val));   // (keeps the load alive)
#endif
}

template <typename T, int FM>
static void launch_dt_pass_t(const DtTask* tasks, int ntasks, const DtMap* maps, const FoldJob* folds, const unsigned long long* foldx, const float* biasw,
                             size_t lds, int nt, hipStream_t s) {
  static LdsOptIn optin;   // one per instantiation, per-device state inside
  optin.ensure((const void*)k_dt_pass<T, FM>, lds);
#ifdef PBD_PROBES
  static const bool tracing = getenv("PBD_DT_TRACE") != nullptr;
  if (tracing) { static int seqs[4096]; const int seq = g_dt_trace_seq++; seqs[seq & 4095] = seq; hipMemcpyToSymbolAsync(HIP_SYMBOL(pbd_dt_trace_launch), &seqs[seq & 4095], sizeof(int), 0, hipMemcpyHostToDevice, s); }
#endif
  hipLaunchKernelGGL((k_dt_pass<T, FM>), dim3(ntasks), dim3(nt), lds, s, tasks, maps, folds, biasw, foldx);
}
template <typename T>
static void launch_dt_pass_m(const DtTask* tasks, int ntasks, const DtMap* maps, const FoldJob* folds, const unsigned long long* foldx, const float* biasw,
                             size_t lds, int nt, int fm, hipStream_t s) {
  // fold launches are instantiated for the smallest register-array bound that holds the model's mixture counts
  if (!folds) launch_dt_pass_t<T, 0>(tasks, ntasks, maps, folds, foldx, biasw, lds, nt, s);
  else if (fm <= 1) launch_dt_pass_t<T, 1>(tasks, ntasks, maps, folds, foldx, biasw, lds, nt, s);
  else if (fm <= 4) launch_dt_pass_t<T, 4>(tasks, ntasks, maps, folds, foldx, biasw, lds, nt, s);
  else if (fm <= 6) launch_dt_pass_t<T, 6>(tasks, ntasks, maps, folds, foldx, biasw, lds, nt, s);
  else launch_dt_pass_t<T, PBD_FOLD_MAXMIX>(tasks, ntasks, maps, folds, foldx, biasw, lds, nt, s);
}
// ts = sizeof(T): DistanceTransform<float> / DistanceTransform<double>; folds != nullptr: the tasks are fold blocks of a
// model whose parts have at most fm mixtures
void launch_dt_pass(const DtTask* tasks, int ntasks, const DtMap* maps, const FoldJob* folds, const unsigned long long* foldx, const float* biasw, size_t lds,
                    int ts, int nt, int fm, hipStream_t s) {
  if (ntasks <= 0) return;
  if (ts == 8) launch_dt_pass_m<double>(tasks, ntasks, maps, folds, foldx, biasw, lds, nt, fm, s);
  else launch_dt_pass_m<float>(tasks, ntasks, maps, folds, foldx, biasw, lds, nt, fm, s);
}

// ---------------------------------------------------------------------------
// reduce over child mixtures, one thread per cell of one (level, child part)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_reduce(const ReduceJob* __restrict__ jobs, const ReduceBlock* __restrict__ blocks,
                                                const float* __restrict__ biasw, int correct_ptr) {
  // block -> (job, first cell) comes from a host-built table; the job descriptor (pointers, bias
  // offsets) is staged in LDS once per block instead of being chased through global memory.
  // (reading the descriptor with scalar loads straight from global memory instead: same time, measured)
  __shared__ ReduceJob J;
  const ReduceBlock rb = blocks[blockIdx.x];
  {
    const int* src = (const int*)(jobs + rb.job);
    int* dst = (int*)&J;
    for (int i = threadIdx.x; i < (int)(sizeof(ReduceJob) / 4); i += 256) dst[i] = src[i];
  }
  __syncthreads();
  const unsigned cell = rb.cell0 + threadIdx.x;
  const int H = J.H, W = J.W, L = J.L;
  const unsigned HW = (unsigned)H * W;
  if (cell >= HW) return;
  constexpr int ML = 8;            // parent mixtures handled with all their gathers in flight at once
  if (L <= ML) {
    T acc[ML];
#pragma unroll
    for (int m = 0; m < ML; ++m) acc[m] = (m < L) ? ((GP(T))J.par_in[m])[cell] : (T)0;
    for (int c = 0; c < J.nch; ++c) {
      const ReduceChild& C = J.ch[c];
      const int K = C.K;
      int bi[ML];
      T v[ML];
      if (K == 1) {  // Math::reduceMax K==1 shortcut: copy (Math.hpp:154-158)
        const T sd = ((GP(T))C.sdt)[cell];
#pragma unroll
        for (int m = 0; m < ML; ++m) { bi[m] = 0; v[m] = (m < L) ? sd + biasw[C.bias_off[0] + m] : (T)0; }
      } else {
#pragma unroll
        for (int m = 0; m < ML; ++m) { bi[m] = 0; v[m] = -INFINITY; }
        for (int mm = 0; mm < K; ++mm) {
          const T sd = ((GP(T))C.sdt)[(size_t)mm * HW + cell];
          const int bo = C.bias_off[mm];
#pragma unroll
          for (int m = 0; m < ML; ++m) {
            if (m < L) {
              const T wv = sd + biasw[bo + m];              // DynamicProgram.cpp:139
              if (wv > v[m]) { bi[m] = mm; v[m] = wv; }         // strict >: first max wins
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < ML; ++m) {
        if (m < L) {
          ((GPW(uint8_t))C.ok)[(size_t)m * HW + cell] = (uint8_t)bi[m];          // Ik (:150); Ix / Iy are composed at back-tracking time
          acc[m] = acc[m] + v[m];                                // parent.score += maxv (:156), child order kept
        }
      }
    }
#pragma unroll
    for (int m = 0; m < ML; ++m) if (m < L) ((GPW(T))J.par_out[m])[cell] = acc[m];
    return;
  }
  for (int m = 0; m < L; ++m) {   // generic path (more than 8 parent mixtures)
    T acc = ((GP(T))J.par_in[m])[cell];
    for (int c = 0; c < J.nch; ++c) {
      const ReduceChild& C = J.ch[c];
      const int K = C.K;
      T v;
      int bi = 0;
      if (K == 1) {
        v = ((GP(T))C.sdt)[cell] + biasw[C.bias_off[0] + m];
      } else {
        v = -INFINITY;
        for (int mm = 0; mm < K; ++mm) {
          const T wv = ((GP(T))C.sdt)[(size_t)mm * HW + cell] + biasw[C.bias_off[mm] + m];
          if (wv > v) { bi = mm; v = wv; }
        }
      }
      ((GPW(uint8_t))C.ok)[(size_t)m * HW + cell] = (uint8_t)bi;
      acc = acc + v;
    }
    ((GPW(T))J.par_out[m])[cell] = acc;
  }
}

void launch_reduce(const ReduceJob* jobs, const ReduceBlock* blocks, int nblocks, const float* biasw, int correct_ptr,
                   int ts, hipStream_t s) {
  if (nblocks <= 0) return;
  if (ts == 8) hipLaunchKernelGGL(k_reduce<double>, dim3(nblocks), dim3(256), 0, s, jobs, blocks, biasw, correct_ptr);
  else hipLaunchKernelGGL(k_reduce<float>, dim3(nblocks), dim3(256), 0, s, jobs, blocks, biasw, correct_ptr);
}

// ---------------------------------------------------------------------------
// root: bias + max over root mixtures, threshold, compaction
// ---------------------------------------------------------------------------
// One 256-thread block = 256 consecutive cells of ONE job (level, component): block -> (job, first cell) comes from a host-built
// table, so the job descriptor is wave-uniform (scalar loads; rounds 1-3 had every thread binary-search the job list: nine
// dependent global loads in front of the fold).  FM: register-array bound of the fold (the model's largest mixture count).
template <typename T, int FM>
__global__ __launch_bounds__(256) void k_root(const RootJob* __restrict__ jobs, const ReduceBlock* __restrict__ blocks, double thresh,
                                              int* __restrict__ count, CandRec* __restrict__ rec, int capacity,
                                              const FoldJob* __restrict__ folds, const float* __restrict__ biasw, int rescan,
                                              const uint8_t* __restrict__ nms_mask, const char* __restrict__ rootv_base) {
  const ReduceBlock rb = blocks[blockIdx.x];
  const RootJob& J = jobs[rb.job];
  const unsigned cell = rb.cell0 + threadIdx.x;
  const unsigned HWr = (unsigned)J.H * (unsigned)J.W;
  const bool live = cell < HWr;
  // (the fold keeps every lane of the block: its children's plane pointers travel in vector-register lanes — fold_child_qw —; lanes past the job's last cell
  //  work on that cell again and store the same Ik bytes once more)
  if (!live && (rescan || J.fold < 0)) return;
  T v;
  int bi = 0;
  const T bias = J.bias;                             // `T bias = root.bias(0)[0]` (:165)
  if (rescan) {
    // root tables handed in by the caller (pbd_set_root), or the second pass of the score-map NMS option (nms_mask: the local
    // maxima of every root plane, same element offsets as the rootv planes): only the threshold and the compaction are redone
    v = ((const T*)J.rootv)[cell];
    const bool keep = !nms_mask || nms_mask[(size_t)((const char*)J.rootv - rootv_base) / sizeof(T) + cell] != 0;
    if ((double)v > thresh && keep) {
      const int idx = atomicAdd(count, 1);
      if (idx < capacity) { CandRec r; r.level = J.level; r.comp = J.comp; r.y = cell / J.W; r.x = cell - r.y * J.W; rec[idx] = r; }
    }
    return;
  }
  const int K = J.K;
  if (J.fold >= 0) {
    // fold mode: the root's accumulated score is built here from its raw responses and its children's messages
    constexpr int M = FM;
    T acc[1][M];
    const unsigned long long c0qw = fold_child_qw(&folds[J.fold].ch[0]);
    const unsigned cellc = live ? cell : HWr - 1u;
    const unsigned offs[1] = {cellc};
#pragma unroll
    for (int m = 0; m < M; ++m) acc[0][m] = ((GP(T))J.score[m])[cellc];      // (entries beyond K repeat mixture K - 1: plan_frame)
    fold_children<T, M, 1>(folds + J.fold, biasw, offs, HWr, K, acc, c0qw, folds[J.fold].nch);
    if (!live) return;
    if (K == 1) {
      v = acc[0][0] + bias;
    } else {
      v = -INFINITY;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (m < K) {
          const T wv = acc[0][m] + bias;                   // DynamicProgram.cpp:169
          if (wv > v) { bi = m; v = wv; }
        }
      }
    }
  } else if (K == 1) {
    v = ((const T*)J.score[0])[cell] + bias;
  } else {
    v = -INFINITY;
    for (int m = 0; m < K; ++m) {
      const T wv = ((const T*)J.score[m])[cell] + bias;  // DynamicProgram.cpp:169
      if (wv > v) { bi = m; v = wv; }
    }
  }
  ((T*)J.rootv)[cell] = v;
  J.rooti[cell] = bi;
  if ((double)v > thresh) {  // :208 strict >
    const int idx = atomicAdd(count, 1);
    if (idx < capacity) {
      CandRec r;
      r.level = J.level; r.comp = J.comp; r.y = cell / J.W; r.x = cell - r.y * J.W;
      rec[idx] = r;
    }
  }
}

template <typename T>
static void launch_root_t(const RootJob* jobs, const ReduceBlock* blocks, int nblocks, double thresh, int* count, CandRec* rec,
                          int capacity, const FoldJob* folds, const float* biasw, int rescan, int fm, const uint8_t* nms_mask,
                          const char* rootv_base, hipStream_t s) {
  const dim3 g(nblocks), b(256);
  if (fm <= 1) hipLaunchKernelGGL((k_root<T, 1>), g, b, 0, s, jobs, blocks, thresh, count, rec, capacity, folds, biasw, rescan, nms_mask, rootv_base);
  else if (fm <= 4) hipLaunchKernelGGL((k_root<T, 4>), g, b, 0, s, jobs, blocks, thresh, count, rec, capacity, folds, biasw, rescan, nms_mask, rootv_base);
  else if (fm <= 6) hipLaunchKernelGGL((k_root<T, 6>), g, b, 0, s, jobs, blocks, thresh, count, rec, capacity, folds, biasw, rescan, nms_mask, rootv_base);
  else hipLaunchKernelGGL((k_root<T, PBD_FOLD_MAXMIX>), g, b, 0, s, jobs, blocks, thresh, count, rec, capacity, folds, biasw, rescan, nms_mask, rootv_base);
}
// fm: largest mixture count of a part (only the fold reads it); blocks: one entry per 256 cells of a job
void launch_root(const RootJob* jobs, const ReduceBlock* blocks, int nblocks, double thresh, int* count, CandRec* rec,
                 int capacity, int ts, const FoldJob* folds, const float* biasw, int rescan, int fm, const uint8_t* nms_mask,
                 const char* rootv_base, hipStream_t s) {
  if (nblocks <= 0) return;
  if (ts == 8) launch_root_t<double>(jobs, blocks, nblocks, thresh, count, rec, capacity, folds, biasw, rescan, fm, nms_mask, rootv_base, s);
  else launch_root_t<float>(jobs, blocks, nblocks, thresh, count, rec, capacity, folds, biasw, rescan, fm, nms_mask, rootv_base, s);
}

// ---------------------------------------------------------------------------
// backtrack (argmin, :219-245): one 64-lane block per candidate, one lane per part, depth by depth — a part's
// location needs only its parent's, so the 26-part skeleton takes 9 dependent steps instead of 25.
// The composed pointer planes Ix/Iy of the reference are not materialised by the DP: for the winning
// child mixture mm = Ik(parent location) the lane reads the DT's own pointer planes of (part, mm),
//   reference composition (DistanceTransform.hpp:233-244):  x = Ix(py, px),  y = Iy(py, x)
//   dt_correct_ptr:                                         y = Iy(py, px),  x = Ix(y, px)
// record = head | boxes[max_parts][4] | locs[max_parts][3]
// ---------------------------------------------------------------------------
#define BT_MAXP 256   // parts per component held in LDS
template <typename T>
__global__ __launch_bounds__(64) void k_backtrack(const int* __restrict__ count, const CandRec* __restrict__ rec,
                                                  int capacity, const BackLevel* __restrict__ back, int ncomp,
                                                  const int* __restrict__ parent, const int* __restrict__ plane0,
                                                  const int* __restrict__ nparts, int max_parts, int kh,
                                                  char* __restrict__ out, size_t out_stride,
                                                  const int* __restrict__ flat, const int* __restrict__ depth, int max_depth,
                                                  int nflat, const unsigned long long* __restrict__ scr_base,
                                                  const int16_t* __restrict__ ixs, const int16_t* __restrict__ iys,
                                                  int correct_ptr, const int16_t* __restrict__ extx,
                                                  const int16_t* __restrict__ exty, const unsigned long long* __restrict__ ext_base,
                                                  int* __restrict__ count_out) {
  __shared__ int lx[BT_MAXP], ly[BT_MAXP], lm[BT_MAXP];
  const int lane = threadIdx.x;
  const int n = min(*count, capacity);
  // count_out != null: `out` and count_out are the handle's PINNED HOST buffers (device-mapped): the records and the count go straight to
  // the host — no copy nodes behind the kernel (two engine copies + their hand-overs cost a single frame ~40 us of its 64 us argmin stage)
  if (count_out && blockIdx.x == 0 && lane == 0) *count_out = *count;
  // grid-stride over the candidates: the launch is sized for the chip, not for the capacity (the count is only known on the
  // device; one block per record of a 32 768-record buffer meant ~30 000 blocks that exit at once, every frame)
  for (int idx = blockIdx.x; idx < n; idx += gridDim.x) {
  __syncthreads();                                   // the previous candidate's lx / ly / lm are done with
  const CandRec r = rec[idx];
  const BackLevel B = back[r.level * ncomp + r.comp];
  const size_t HW = (size_t)B.H * B.W;
  char* o = out + (size_t)idx * out_stride;
  pbd_candidate_head* head = (pbd_candidate_head*)o;
  int32_t* boxes = (int32_t*)(o + sizeof(pbd_candidate_head));
  int32_t* locs = boxes + (size_t)max_parts * 4;
  const int np = nparts[r.comp];
  if (lane == 0) {
    head->score = (float)((const T*)B.rootv)[(size_t)r.y * B.W + r.x];   // Candidate::addPart(Rect, float), Candidate.hpp:72
    head->component = r.comp;
    head->level = r.level;
    head->nparts = np;
    lx[0] = r.x; ly[0] = r.y; lm[0] = B.rooti[(size_t)r.y * B.W + r.x];
  }
  __syncthreads();
  for (int d = 1; d <= max_depth; ++d) {
    for (int p = lane; p < np; p += 64) {
      if (depth[r.comp * max_parts + p] != d) continue;
      const int par = parent[r.comp * max_parts + p];
      const int px = lx[par], py = ly[par], pm = lm[par];
      const size_t off = (size_t)py * B.W + px;
      const int mm = B.pk[(size_t)(plane0[r.comp * max_parts + p] + pm) * HW + off];           // Ik
      int x, y;
      if (extx) {   // tables handed in by the caller (pbd_set_dp_pointers): Ix / Iy are stored composed, per (part, parent mixture)
        const size_t eo = (size_t)ext_base[r.level * ncomp + r.comp] + (size_t)(plane0[r.comp * max_parts + p] + pm) * HW + off;
        x = extx[eo]; y = exty[eo];
      } else {
        const size_t so = (size_t)scr_base[(size_t)r.level * nflat + flat[r.comp * max_parts + p]] + (size_t)mm * HW;
        if (!correct_ptr) { x = ixs[so + off]; y = iys[so + (size_t)py * B.W + x]; }
        else { y = iys[so + off]; x = ixs[so + (size_t)y * B.W + px]; }
      }
      lx[p] = x; ly[p] = y; lm[p] = mm;
    }
    __syncthreads();
  }
  const T scale = B.scale;                             // `T scale = scales[n]` (:198): Point * T rounds with cvRound
  const int sz = t_round((T)kh * scale);               // Point(xsize,ysize)*scale
  for (int p = lane; p < max_parts; p += 64) {
    if (p < np) {
      const int x = lx[p], y = ly[p];
      locs[p * 3] = x; locs[p * 3 + 1] = y; locs[p * 3 + 2] = lm[p];
      const int x1 = t_round((T)(x - 1) * scale), y1 = t_round((T)(y - 1) * scale);
      const int x2 = x1 + sz - 1, y2 = y1 + sz - 1;
      boxes[p * 4] = min(x1, x2); boxes[p * 4 + 1] = min(y1, y2);
      boxes[p * 4 + 2] = max(x1, x2) - min(x1, x2); boxes[p * 4 + 3] = max(y1, y2) - min(y1, y2);
    } else {                                           // components with fewer parts: zero padding
      for (int k = 0; k < 4; ++k) boxes[p * 4 + k] = 0;
      for (int k = 0; k < 3; ++k) locs[p * 3 + k] = 0;
    }
  }
  }
}

void launch_backtrack(const int* count, const CandRec* rec, int capacity, const BackLevel* back, int ncomp,
                      const int* parent, const int* plane0, const int* nparts, int max_parts, int kh, char* out,
                      size_t out_stride, int ts, const int* flat, const int* depth, int max_depth, int nflat,
                      const unsigned long long* scr_base, const int16_t* ix, const int16_t* iy, int correct_ptr,
                      const int16_t* extx, const int16_t* exty, const unsigned long long* ext_base, int* count_out, hipStream_t s) {
  const int nblk = std::min(capacity, 2048);   // 8 blocks of one wavefront per CU; more candidates than that are taken in further sweeps
  if (ts == 8) hipLaunchKernelGGL(k_backtrack<double>, dim3(nblk), dim3(64), 0, s, count, rec, capacity, back, ncomp, parent, plane0, nparts, max_parts, kh, out, out_stride, flat, depth, max_depth, nflat, scr_base, ix, iy, correct_ptr, extx, exty, ext_base, count_out);
  else hipLaunchKernelGGL(k_backtrack<float>, dim3(nblk), dim3(64), 0, s, count, rec, capacity, back, ncomp, parent, plane0, nparts, max_parts, kh, out, out_stride, flat, depth, max_depth, nflat, scr_base, ix, iy, correct_ptr, extx, exty, ext_base, count_out);
}

// ---------------------------------------------------------------------------
// Neubeck & Van Gool block NMS on a score map (reference src/nms.cpp:84-129; dead
// code there, offered as an optional pre-filter).  One thread per (sz+1)^2 block.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void nms_block(const T* __restrict__ src, int M, int N, int sz, uint8_t* __restrict__ dst, int b) {
  const int nbx = (N + sz) / (sz + 1), nby = (M + sz) / (sz + 1);
  if (b >= nbx * nby) return;
  const int m = (b / nbx) * (sz + 1), n = (b % nbx) * (sz + 1);
  const int i1 = min(m + sz + 1, M), j1 = min(n + sz + 1, N);
  double vcmax = -1.7976931348623157e308;
  int ci = m, cj = n;
  for (int i = m; i < i1; ++i)
    for (int j = n; j < j1; ++j) {
      const double v = src[(size_t)i * N + j];
      if (v > vcmax) { vcmax = v; ci = i; cj = j; }
    }
  const int in0 = max(ci - sz, 0), in1 = min(ci + sz + 1, M);
  const int jn0 = max(cj - sz, 0), jn1 = min(cj + sz + 1, N);
  const int is0 = m - in0, is1 = min(m - in0 + sz + 1, in1 - in0);
  const int js0 = n - jn0, js1 = min(n - jn0 + sz + 1, jn1 - jn0);
  double vnmax = -1.7976931348623157e308;
  bool any = false;
  for (int i = in0; i < in1; ++i)
    for (int j = jn0; j < jn1; ++j) {
      const int li = i - in0, lj = j - jn0;
      if (li >= is0 && li < is1 && lj >= js0 && lj < js1) continue;
      any = true;
      const double v = src[(size_t)i * N + j];
      if (v > vnmax) vnmax = v;
    }
  if (!any) vnmax = 0;
  if (vcmax > vnmax) dst[(size_t)ci * N + cj] = 255;
}
__global__ __launch_bounds__(64) void k_nms_map(const float* __restrict__ src, int M, int N, int sz,
                                                uint8_t* __restrict__ dst) {
  nms_block<float>(src, M, N, sz, dst, blockIdx.x * 64 + threadIdx.x);
}
// the same on the resident root-score planes of a frame (or batch): blockIdx.y = root job (level, component); the mask plane of
// a job sits at the element offset of its rootv plane
template <typename T>
__global__ __launch_bounds__(64) void k_nms_roots(const RootJob* __restrict__ jobs, const char* __restrict__ rootv_base, int sz,
                                                  uint8_t* __restrict__ mask) {
  const RootJob& J = jobs[blockIdx.y];
  nms_block<T>((const T*)J.rootv, J.H, J.W, sz, mask + (size_t)((const char*)J.rootv - rootv_base) / sizeof(T), blockIdx.x * 64 + threadIdx.x);
}
void launch_nms_roots(const RootJob* jobs, int njobs, unsigned maxcells, const char* rootv_base, int ts, int sz, uint8_t* mask, hipStream_t s) {
  if (njobs <= 0 || maxcells == 0) return;
  // blocks of a plane <= cells of it (sz >= 1: fewer); one thread per NMS block, lanes past the job's blocks return
  const dim3 grid((maxcells + 63) / 64, njobs);
  if (ts == 8) hipLaunchKernelGGL(k_nms_roots<double>, grid, dim3(64), 0, s, jobs, rootv_base, sz, mask);
  else hipLaunchKernelGGL(k_nms_roots<float>, grid, dim3(64), 0, s, jobs, rootv_base, sz, mask);
}

void launch_nms_map(const float* src, int rows, int cols, int sz, uint8_t* dst, hipStream_t s) {
  const int nb = ((cols + sz) / (sz + 1)) * ((rows + sz) / (sz + 1));
  hipMemsetAsync(dst, 0, (size_t)rows * cols, s);
  hipLaunchKernelGGL(k_nms_map, dim3((nb + 63) / 64), dim3(64), 0, s, src, rows, cols, sz, dst);
}
