// k_dp.hip — DynamicProgram<T>::min / argmin on the GPU.
// Reference: src/DynamicProgram.cpp:66-255, include/DistanceTransform.hpp:151-245,
// include/Math.hpp:108-185.
//
// k_dt_pass   one 1-D generalised distance transform pass (Felzenszwalb &
//             Huttenlocher upper envelope, DistanceTransform.hpp:151-182).
//             One lane per score line, 64 lines (possibly from several maps of
//             the same level) per wavefront.  The envelope is built with the
//             reference's exact arithmetic: intersection in fp64, narrowed to
//             fp32 (`T s = f(...)`, :161), `s <= z[k]` pops (:162), read-out
//             `z[k+1] < os` with the int promoted to float (:174) and the value
//             evaluated in fp64 (:175).  Stack (v, z) and the line live in LDS
//             (the y-values of stack entries overwrite consumed line entries in
//             place).  Lines are read coalesced (line-contiguous input) and the
//             result is written TRANSPOSED (element q of line i at q*nlines+i),
//             so the x pass (rows) feeds the y pass (columns) line-contiguously
//             and the y pass lands in natural row-major layout again, every
//             global access coalesced across the 64 lanes.
// k_reduce    Math::reduceMax + reducePickIndex over the child mixtures for
//             every parent mixture, the reference's pointer composition
//             Iy'(m,n) = Iy(m, Ix(m,n)) (DistanceTransform.hpp:233-244), and the
//             in-order accumulation into the parent score (DynamicProgram.cpp:134-156).
// k_root      root bias + reduceMax (:163-171), strict threshold (:208) and
//             compaction of the hits.
// k_backtrack argmin (:219-245): one lane per candidate walks the part tree.
#include <type_traits>
#include "pbd_internal.hpp"

// debug: per-phase timestamps (100 MHz wall clock) of block 0 of the last k_dt_pass launch
#ifdef PBD_PROBES
__device__ unsigned long long pbd_dt_dbg[8];
#define DT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) pbd_dt_dbg[i] = wall_clock64(); } while (0)
void dt_debug_read(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(pbd_dt_dbg), sizeof(unsigned long long) * 8); }
#else
#define DT_STAMP(i) do { } while (0)
void dt_debug_read(unsigned long long* out) { for (int i = 0; i < 8; ++i) out[i] = 0; }
#endif

// LDS per block: 64 line pointers + 64 stack sizes + per line {(Y, Z) : float2[S]; V : u8[S] (S <= 256) or
// u16[S]} + per map touched by the block a table of exact reciprocals 1/(2a*dx), dx < len (double[S]).
// The envelope scan is VALU-issue bound with one lane per line, and the lines resident on a CU are
// bounded by these bytes: every byte saved per element is more active lanes per wavefront.
size_t dt_lds_bytes(int stride, int lpb, int nmb, int ts) {   // ts = sizeof(T): (Y, Z) is float2 or double2
  return (size_t)lpb * stride * (2 * ts + (stride <= 256 ? 1 : 2)) + 64 * (8 + 8 + 4 + 4) + (size_t)nmb * stride * 8 + 16;
}
template <typename T> struct Pair;                       // (y, z) of one stack entry: one LDS word pair
template <> struct Pair<float> { typedef float2 type; };
template <> struct Pair<double> { typedef double2 type; };

// Envelope scan of one line (DistanceTransform.hpp:156-170), one lane per line.
//
// Intersection of the parabolas rooted at x0 < x1 (Quadratic::operator()(x0,x1,y0,y1), :98-100),
// narrowed to T like `T s = f(...)` at :161:
//   num = ((y1 - y0) - b*(x1-x0)) + a*(x1^2 - x0^2)      (same fp64 operations, same order)
//   s   = (float)(num / den),  den = (2a)*(x1-x0)
// EXACT = false: the fp64 division (14 dependent instructions, four of them quarter rate) is
// replaced by ONE multiplication with the correctly rounded reciprocal r = RN(1/den) from the per-map
// table: q1 = RN(num * r) carries two roundings (r, the product), so |q1 - num/den| <= (2u + u^2)|num/den|
// and q1 lies within 3 ulp of the reference's RN(num/den).  (float)q1 can therefore differ from
// (float)RN(num/den) only if a float rounding boundary (a double whose low 29 mantissa bits are
// 0x10000000) lies within 3 ulp of q1 — low 29 bits in 0x0FFFFFFC..0x10000004 are flagged, one ulp of
// slack — or the value leaves the normal float range.  Those cases (~2e-8 of all evaluations) set a
// sticky flag and the whole line is redone with EXACT = true (true IEEE division), so the result is
// always bit-identical to the reference's.
//
// The reference's nested loops (for q { while (pop) }) are flattened into a state machine doing
// exactly one intersection per iteration: lanes never wait for the slowest lane's pop count and
// every line sees the reference's sequence of intersections / `s <= z[k]` tests in order.  A
// single in-order wave exposes every latency and is instruction-issue bound, so the body is
// branch-free, keeps the stack top and the entry below it in registers, prefetches the entry two
// below, the reciprocal for a pop and the next line element at the top of the iteration (consumed
// at its end), stores (y,z) of an entry as one 8-byte LDS word, and issues the push stores
// unconditionally to slot k+1 (dead when the step pops).  The y of stack entry k overwrites the
// consumed line element k in place (k <= q).
template <bool EXACT, typename VT, typename T>
__device__ __forceinline__ bool dt_envelope(typename Pair<T>::type* __restrict__ YZl, VT* __restrict__ Vl,
                                            const double* __restrict__ Rl, int len, double a, double b, int* kout) {
  const double twoa = 2 * a;
  const double r1 = Rl[1 < len ? 1 : 0];
  int k = 0, q = 1, vk = 0, nv = 0;
  T zk = -INFINITY, nz = -INFINITY;
  double yk = (double)YZl[0].x, ny = 0.0;
  double r_top = r1;
  unsigned suspect = 0;             // sticky: a quotient landed next to a float rounding boundary
  Vl[0] = 0;
  YZl[0].y = -INFINITY;
  T yq_f = YZl[min(1, len - 1)].x;
  while (q < len) {
    // prefetches (addresses known now, values used after the arithmetic below)
    const int k2 = max(k - 2, 0);
    const int pv = Vl[k2];
    const typename Pair<T>::type pyz = YZl[k2];
    const double r_nxt = Rl[max(q - nv, 0)];   // reciprocal for the entry below the top (used if this step pops)
    const T ynext_f = YZl[min(q + 1, len - 1)].x;
    // intersection with the stack top
    const int dx = q - vk;
    const double yq = (double)yq_f;
    const double dxd = (double)dx;
    const double num = ((yq - yk) - b * dxd) + a * (double)__mul24(dx, q + vk);   // x1^2 - x0^2 < 2^31, operands < 2^16
    const double den = twoa * dxd;
    double q1;
    if (EXACT) {
      q1 = num / den;
    } else {
      q1 = num * r_top;                 // within 3 ulp of RN(num/den): |r - 1/den| <= u/den, one more rounding
      const unsigned long long bits = (unsigned long long)__double_as_longlong(q1);
      const unsigned lo29 = (unsigned)bits & 0x1FFFFFFFu;
      const unsigned ex = (unsigned)(bits >> 52) & 0x7FFu;
      suspect = (((lo29 - 0x0FFFFFFCu) <= 8u) | ((ex - 897u) > 252u)) ? 1u : suspect;   // one v_cndmask
    }
    const T s = (T)q1;                          // `T s = f(...)` (:161): narrowed for float, kept for double
    const bool pop = (s <= zk) && (k > 0);  // :162
    // push stores (:166-169); slot k+1 is dead if this step pops
    Vl[k + 1] = (VT)q;
    { typename Pair<T>::type e; e.x = yq_f; e.y = s; YZl[k + 1] = e; }
    // state update, selects only
    const int vk_o = vk; const double yk_o = yk; const T zk_o = zk;
    k = pop ? k - 1 : k + 1;
    vk = pop ? nv : q;
    yk = pop ? ny : yq;
    zk = pop ? nz : s;
    r_top = pop ? r_nxt : r1;
    nv = pop ? pv : vk_o;
    ny = pop ? (double)pyz.x : yk_o;
    nz = pop ? pyz.y : zk_o;
    yq_f = pop ? yq_f : ynext_f;
    q = pop ? q : q + 1;
  }
  *kout = k;
  return suspect != 0;
}

// Cooperative envelope scan: LPL (2, 4, 8 or 16) adjacent lanes share one line.  Lane j of the group holds stack
// entry k-j and evaluates the intersection of ITS entry with the current element q — exactly the value the
// reference computes when its pop loop reaches that entry (`s = f(v[k], q, ...)` after j pops, :161-165).
// pop_j = (s_j <= z[k-j]) && (k-j > 0); the reference pops while that holds, so the number of pops is
// p = index of the first lane whose test fails.  p < LPL: lane p pushes q with its own s at slot k-p+1 and
// the element is consumed — a push with up to LPL-1 pops costs ONE iteration; p == LPL: all LPL entries pop
// and the same element meets the next LPL entries.  Same comparisons, same values, same results, about
// one iteration per element instead of 1.5, and the extra lanes are lanes that LDS capacity leaves idle
// anyway (a block holds 14-16 lines of 158 elements).  Loop state is just (k, q): entries are re-read
// from the LDS stack every iteration (a wave's LDS operations are ordered; the fence keeps the compiler
// from hoisting a lane's read above another lane's push).
template <bool EXACT, int LPL, typename VT, typename T>
__device__ __forceinline__ bool dt_envelope_m(typename Pair<T>::type* __restrict__ YZl, VT* __restrict__ Vl,
                                              const double* __restrict__ Rl, int len, double a, double b, int j,
                                              int lane, int* kout) {
  typedef typename Pair<T>::type P2;
  const double twoa = 2 * a;
  const int gshift = lane & ~(LPL - 1);
  int k = 0, q = 1;
  unsigned suspect = 0;
  if (j == 0) { Vl[0] = 0; YZl[0].y = -INFINITY; }
  while (q < len) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int e = k - j, ec = max(e, 0);
    const int v = Vl[ec];
    const P2 yz = YZl[ec];
    const T yq_f = YZl[q].x;
    const int dx = q - v;
    const double r = Rl[dx];
    const double dxd = (double)dx;
    const double num = (((double)yq_f - (double)yz.x) - b * dxd) + a * (double)__mul24(dx, q + v);
    const double den = twoa * dxd;
    double q1;
    if (EXACT) {
      q1 = num / den;
    } else {
      q1 = num * r;
      const unsigned long long bits = (unsigned long long)__double_as_longlong(q1);
      const unsigned lo29 = (unsigned)bits & 0x1FFFFFFFu;
      const unsigned ex = (unsigned)(bits >> 52) & 0x7FFu;
      suspect = (((lo29 - 0x0FFFFFFCu) <= 8u) | ((ex - 897u) > 252u)) ? 1u : suspect;
    }
    const T s = (T)q1;
    // pop = (s <= z[k-j]) && (k-j > 0), :162 for the entry this lane holds; two ballots of plain compares
    // ANDed as scalars (a ballot of the && goes through a select and a second compare)
    const unsigned long long pm = __builtin_amdgcn_ballot_w64(s <= yz.y) & __builtin_amdgcn_ballot_w64(e > 0);
    const unsigned gm = (unsigned)(pm >> gshift) & ((1u << LPL) - 1u);
    const int p = __builtin_ctz(~gm);                                     // pops before the first failing test: 0..LPL
    const bool push = p < LPL;
    const int kn = push ? k - p + 1 : k - LPL;
    if (push && j == p) {                                                 // :166-169 by the lane whose test failed
      Vl[kn] = (VT)q;
      P2 en; en.x = yq_f; en.y = s;
      YZl[kn] = en;
    }
    k = kn;
    q = push ? q + 1 : q;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  *kout = k;
  return suspect != 0;
}

// One block = one wavefront = up to g.lpb lines of one group (lpb chosen per group so that every
// block of the launch fits the same LDS budget: long lines -> fewer lines per block -> many more
// blocks, so a whole pass is resident at once and all 4 SIMDs of every CU carry chains).
template <typename T, typename VT>
__device__ __forceinline__ void dt_block(char* smem, const DtTask& t, const DtGroup& g, const DtMap* __restrict__ maps) {
  const int lane = threadIdx.x;
  const int len = g.len, S = g.stride, lpb = g.lpb;
  typedef typename Pair<T>::type P2;
  const T** lptr = (const T**)smem;   // [64] source pointer of each line of this block
  int16_t** pptr = (int16_t**)(smem + 64 * 8);  // [64] pointer-output base of each line
  int* pstr = (int*)(smem + 64 * 16);         // [64] pointer-output element stride of each line
  int* Ksz = (int*)(smem + 64 * 20);          // [64] final stack size of each line
  double* R = (double*)(smem + 64 * 24);      // [nmb][S] 1/(2a*dx) per map of this block
  const int total = g.nmaps * g.nlines;
  const int nl = min(lpb, total - t.g0);
  const int m_first = t.g0 / g.nlines, m_last = (t.g0 + nl - 1) / g.nlines;
  const int nmb = m_last - m_first + 1;
  P2* YZ = (P2*)(R + g.nmb * S);      // [lpb][S] .x: line values, then y of stack entries (in place); .y: z[k]
  VT* V = (VT*)(YZ + lpb * S);                // [lpb][S] v[k]
  if (lane < nl) {
    const int gi = t.g0 + lane;
    const int mi = gi / g.nlines, li = gi - mi * g.nlines;
    const DtMap& mp0 = maps[g.map0 + mi];
    lptr[lane] = (const T*)mp0.src + (size_t)li * len;
    // pointers: transposed like dst (y pass -> natural layout) or natural (x pass)
    const bool nat = mp0.ptr_natural != 0;
    pptr[lane] = mp0.ptr + (nat ? (size_t)li * len : (size_t)li);
    pstr[lane] = nat ? 1 : g.nlines;
  }
  // reciprocal tables: one IEEE division per (map, dx), spread over the 64 lanes
  for (int ms = 0; ms < nmb; ++ms) {
    const double a = maps[g.map0 + m_first + ms].a;
    for (int dx = lane; dx < len; dx += 64) R[ms * S + dx] = 1.0 / ((2 * a) * (double)dx);
  }
  __syncthreads();
  DT_STAMP(1);
  // coalesced load of the nl lines.  Batches of LB independent loads are issued before the first
  // wait (addresses are clamped instead of predicated: a predicated load makes hipcc branch and
  // wait per element, which serialises one full memory round trip per 256 B).
  {
    const int CH = (len + 63) >> 6;          // 64-element chunks per line
    const int nch = nl * CH;
    // chunk -> (line, chunk of line) by a reciprocal multiply: an integer division per load and per store
    // costs more VALU time than the loads take (c * CH < 2^20 here: c < 64 * CH, CH <= 512)
    const unsigned inv = (1u << 20) / (unsigned)CH + 1u;
    constexpr int LB = 36;                   // loads in flight per lane (one round trip covers a whole 11-12 line block)
    for (int c0 = 0; c0 < nch; c0 += LB) {
      T r[LB];
#pragma unroll
      for (int j = 0; j < LB; ++j) {
        const int c = min(c0 + j, nch - 1);
        const int i = (int)(((unsigned)c * inv) >> 20);
        const int q = min((c - i * CH) * 64 + lane, len - 1);
        r[j] = lptr[i][q];
      }
#pragma unroll
      for (int j = 0; j < LB; ++j) {
        const int c = c0 + j, cc = min(c, nch - 1);
        const int i = (int)(((unsigned)cc * inv) >> 20);
        const int q = (cc - i * CH) * 64 + lane;
        if (c < nch && q < len) YZ[i * S + q].x = r[j];
      }
    }
  }
  __syncthreads();
  DT_STAMP(2);

  // ---- build the upper envelope (DistanceTransform.hpp:156-170) ----
  // lanes per line: long lines leave most lanes of the wave without a line (LDS capacity), so 4 or 2 lanes
  // share a line (dt_envelope_m); short lines fill the wave with one lane per line (dt_envelope).
  const int lpl = g.lpb <= 4 ? 16 : g.lpb <= 8 ? 8 : g.lpb <= 16 ? 4 : (g.lpb <= 32 ? 2 : 1);
  auto coop = [&](auto LPLc) {
    constexpr int LPL = decltype(LPLc)::value;
    const int line = lane / LPL, j = lane % LPL;
    if (line < nl) {
      const int gi = t.g0 + line;
      const int mi = gi / g.nlines;
      const DtMap mp = maps[g.map0 + mi];
      const double* Rl = R + (mi - m_first) * S;
      P2* YZl = YZ + line * S;
      VT* Vl = V + line * S;
      int k;
      if constexpr (sizeof(T) == 8) {
        dt_envelope_m<true, LPL, VT, T>(YZl, Vl, Rl, len, mp.a, mp.b, j, lane, &k);
      } else {
        const bool sus = dt_envelope_m<false, LPL, VT, T>(YZl, Vl, Rl, len, mp.a, mp.b, j, lane, &k);
        const unsigned gm = (unsigned)(__ballot(sus) >> (lane & ~(LPL - 1))) & ((1u << LPL) - 1u);
        if (gm) {   // a quotient of this line landed next to a float rounding boundary: redo it with true divisions
          const T* src = lptr[line];
          for (int q = j; q < len; q += LPL) YZl[q].x = src[q];
          dt_envelope_m<true, LPL, VT, T>(YZl, Vl, Rl, len, mp.a, mp.b, j, lane, &k);
        }
      }
      if (j == 0) { YZl[k + 1].y = INFINITY; Ksz[line] = k; }
    }
  };
  if (lpl == 16) coop(std::integral_constant<int, 16>());
  else if (lpl == 8) coop(std::integral_constant<int, 8>());
  else if (lpl == 4) coop(std::integral_constant<int, 4>());
  else if (lpl == 2) coop(std::integral_constant<int, 2>());
  else if (lane < nl) {
    const int gi = t.g0 + lane;
    const int mi = gi / g.nlines;
    const DtMap mp = maps[g.map0 + mi];
    const double* Rl = R + (mi - m_first) * S;
    P2* YZl = YZ + lane * S;
    VT* Vl = V + lane * S;
    int k;
    if constexpr (sizeof(T) == 8) {
      // DistanceTransform<double>: s is not narrowed, so every intersection takes the IEEE division
      dt_envelope<true, VT, T>(YZl, Vl, Rl, len, mp.a, mp.b, &k);
    } else if (dt_envelope<false, VT, T>(YZl, Vl, Rl, len, mp.a, mp.b, &k)) {
      // a quotient landed within 1 ulp of a float rounding boundary: redo this line with true divisions
      const T* src = lptr[lane];
      for (int q = 0; q < len; ++q) YZl[q].x = src[q];
      dt_envelope<true, VT, T>(YZl, Vl, Rl, len, mp.a, mp.b, &k);
    }
    YZl[k + 1].y = INFINITY;
    Ksz[lane] = k;
  }
  __syncthreads();
  DT_STAMP(3);

  // ---- read out (:172-178) ----
  // Output q of a line depends only on the finished stack, so the 64/lpb idle lane groups share a
  // line: sub-range r of the outputs starts from a binary search for its first stack entry.
  // Scores go out transposed (lanes of one sub-range = consecutive lines -> coalesced); pointers go
  // straight to their plane: transposed like the scores in the y pass, natural (2-byte runs per lane,
  // merged in L2) in the x pass — no LDS staging, that space holds more lines instead.
  {
    const int nsub = 64 / lpb;
    const int line = lane % lpb, sub = lane / lpb;
    if (line < nl && sub < nsub) {
      const int gi = t.g0 + line;
      const int mi = gi / g.nlines, li = gi - mi * g.nlines;
      const DtMap mp = maps[g.map0 + mi];
      const double a = mp.a, b = mp.b;
      const P2* YZl = YZ + line * S;
      const VT* Vl = V + line * S;
      int16_t* pp = pptr[line];
      const int pst = pstr[line];
      const int K = Ksz[line];
      const int chunk = (len + nsub - 1) / nsub;
      const int q0 = sub * chunk, q1 = min(len, q0 + chunk);
      if (q0 < q1) {
        int os = mp.os + q0;
        // first k with !(z[k+1] < os): z is strictly increasing, z[K+1] = +inf
        int lo = 0, hi = K;
        const T f0 = (T)os;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (YZl[mid + 1].y < f0) lo = mid + 1; else hi = mid;
        }
        int k = lo;
        int vk = Vl[k];
        T yk = YZl[k].x;
        // the piece after the current one is kept in registers (its y and z are one LDS word), so stepping to it
        // costs no LDS round trip on the spot: the read of the piece after THAT overlaps this output's arithmetic
        P2 nyz = YZl[k + 1];                     // (y[k+1], z[k+1]); z[K+1] = +inf ends the walk, slots up to K+2 exist
        int nv = Vl[k + 1];
        const int nlines = g.nlines;
        T* dp = (T*)mp.dst + li + (size_t)q0 * nlines;      // running output pointers: no 64-bit multiply per element
        int16_t* ppq = pp + (size_t)q0 * pst;
        for (int q = q0; q < q1; ++q) {
          const T fos = (T)os;                   // `z[k+1] < os`: int promoted to T (:174)
          while (nyz.y < fos) { k++; vk = nv; yk = nyz.x; nyz = YZl[k + 1]; nv = Vl[k + 1]; }
          const int d = os - vk;
          *dp = (T)(a * (double)__mul24(d, d) + b * (double)d + (double)yk);   // |d| < 2^15
          *ppq = (int16_t)vk;
          dp += nlines; ppq += pst;
          os++;
        }
      }
    }
  }
  DT_STAMP(4);
  DT_STAMP(5);
}

template <typename T>
__global__ __launch_bounds__(64) void k_dt_pass(const DtTask* __restrict__ tasks, const DtGroup* __restrict__ groups,
                                                const DtMap* __restrict__ maps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  DT_STAMP(0);
  const DtTask t = tasks[blockIdx.x];
  const DtGroup g = groups[t.group];
  if (g.stride <= 256) dt_block<T, unsigned char>(smem, t, g, maps);    // stack indices < 255 fit a byte
  else dt_block<T, unsigned short>(smem, t, g, maps);
}

template <typename T>
static void launch_dt_pass_t(const DtTask* tasks, int ntasks, const DtGroup* groups, const DtMap* maps, size_t lds,
                             hipStream_t s) {
  static LdsOptIn optin;   // one per instantiation, per-device state inside
  optin.ensure((const void*)k_dt_pass<T>, lds);
  hipLaunchKernelGGL(k_dt_pass<T>, dim3(ntasks), dim3(64), lds, s, tasks, groups, maps);
}
// ts = sizeof(T): DistanceTransform<float> / DistanceTransform<double>
void launch_dt_pass(const DtTask* tasks, int ntasks, const DtGroup* groups, const DtMap* maps, size_t lds, int ts,
                    hipStream_t s) {
  if (ntasks <= 0) return;
  if (ts == 8) launch_dt_pass_t<double>(tasks, ntasks, groups, maps, lds, s);
  else launch_dt_pass_t<float>(tasks, ntasks, groups, maps, lds, s);
}

// ---------------------------------------------------------------------------
// reduce over child mixtures, one thread per cell of one (level, child part)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_reduce(const ReduceJob* __restrict__ jobs, const ReduceBlock* __restrict__ blocks,
                                                const float* __restrict__ biasw, int correct_ptr) {
  // block -> (job, first cell) comes from a host-built table; the job descriptor (pointers, bias
  // offsets) is staged in LDS once per block instead of being chased through global memory.
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
    for (int m = 0; m < ML; ++m) acc[m] = (m < L) ? ((const T*)J.par_in[m])[cell] : (T)0;
    for (int c = 0; c < J.nch; ++c) {
      const ReduceChild& C = J.ch[c];
      const int K = C.K;
      int bi[ML];
      T v[ML];
      if (K == 1) {  // Math::reduceMax K==1 shortcut: copy (Math.hpp:154-158)
        const T sd = ((const T*)C.sdt)[cell];
#pragma unroll
        for (int m = 0; m < ML; ++m) { bi[m] = 0; v[m] = (m < L) ? sd + biasw[C.bias_off[0] + m] : (T)0; }
      } else {
#pragma unroll
        for (int m = 0; m < ML; ++m) { bi[m] = 0; v[m] = -INFINITY; }
        for (int mm = 0; mm < K; ++mm) {
          const T sd = ((const T*)C.sdt)[(size_t)mm * HW + cell];
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
          C.ok[(size_t)m * HW + cell] = (uint8_t)bi[m];          // Ik (:150); Ix / Iy are composed at back-tracking time
          acc[m] = acc[m] + v[m];                                // parent.score += maxv (:156), child order kept
        }
      }
    }
#pragma unroll
    for (int m = 0; m < ML; ++m) if (m < L) ((T*)J.par_out[m])[cell] = acc[m];
    return;
  }
  for (int m = 0; m < L; ++m) {   // generic path (more than 8 parent mixtures)
    T acc = ((const T*)J.par_in[m])[cell];
    for (int c = 0; c < J.nch; ++c) {
      const ReduceChild& C = J.ch[c];
      const int K = C.K;
      T v;
      int bi = 0;
      if (K == 1) {
        v = ((const T*)C.sdt)[cell] + biasw[C.bias_off[0] + m];
      } else {
        v = -INFINITY;
        for (int mm = 0; mm < K; ++mm) {
          const T wv = ((const T*)C.sdt)[(size_t)mm * HW + cell] + biasw[C.bias_off[mm] + m];
          if (wv > v) { bi = mm; v = wv; }
        }
      }
      C.ok[(size_t)m * HW + cell] = (uint8_t)bi;
      acc = acc + v;
    }
    ((T*)J.par_out[m])[cell] = acc;
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
template <typename T>
__global__ __launch_bounds__(256) void k_root(const RootJob* __restrict__ jobs, int njobs, double thresh,
                                              int* __restrict__ count, CandRec* __restrict__ rec, int capacity) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].cell0 <= gid) lo = mid; else hi = mid - 1;
  }
  const RootJob& J = jobs[lo];
  const unsigned cell = gid - J.cell0;
  if (cell >= (unsigned)J.H * J.W) return;
  T v;
  int bi = 0;
  const T bias = J.bias;                             // `T bias = root.bias(0)[0]` (:165)
  if (J.K == 1) {
    v = ((const T*)J.score[0])[cell] + bias;
  } else {
    v = -INFINITY;
    for (int m = 0; m < J.K; ++m) {
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

void launch_root(const RootJob* jobs, int njobs, unsigned total_cells, double thresh, int* count, CandRec* rec,
                 int capacity, int ts, hipStream_t s) {
  if (njobs <= 0 || total_cells == 0) return;
  if (ts == 8) hipLaunchKernelGGL(k_root<double>, dim3((total_cells + 255) / 256), dim3(256), 0, s, jobs, njobs, thresh, count, rec, capacity);
  else hipLaunchKernelGGL(k_root<float>, dim3((total_cells + 255) / 256), dim3(256), 0, s, jobs, njobs, thresh, count, rec, capacity);
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
                                                  int correct_ptr) {
  __shared__ int lx[BT_MAXP], ly[BT_MAXP], lm[BT_MAXP];
  const int idx = blockIdx.x, lane = threadIdx.x;
  const int n = min(*count, capacity);
  if (idx >= n) return;
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
      const size_t so = (size_t)scr_base[(size_t)r.level * nflat + flat[r.comp * max_parts + p]] + (size_t)mm * HW;
      int x, y;
      if (!correct_ptr) { x = ixs[so + off]; y = iys[so + (size_t)py * B.W + x]; }
      else { y = iys[so + off]; x = ixs[so + (size_t)y * B.W + px]; }
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

void launch_backtrack(const int* count, const CandRec* rec, int capacity, const BackLevel* back, int ncomp,
                      const int* parent, const int* plane0, const int* nparts, int max_parts, int kh, char* out,
                      size_t out_stride, int ts, const int* flat, const int* depth, int max_depth, int nflat,
                      const unsigned long long* scr_base, const int16_t* ix, const int16_t* iy, int correct_ptr,
                      hipStream_t s) {
  if (ts == 8) hipLaunchKernelGGL(k_backtrack<double>, dim3(capacity), dim3(64), 0, s, count, rec, capacity, back, ncomp, parent, plane0, nparts, max_parts, kh, out, out_stride, flat, depth, max_depth, nflat, scr_base, ix, iy, correct_ptr);
  else hipLaunchKernelGGL(k_backtrack<float>, dim3(capacity), dim3(64), 0, s, count, rec, capacity, back, ncomp, parent, plane0, nparts, max_parts, kh, out, out_stride, flat, depth, max_depth, nflat, scr_base, ix, iy, correct_ptr);
}

// ---------------------------------------------------------------------------
// Neubeck & Van Gool block NMS on a score map (reference src/nms.cpp:84-129; dead
// code there, offered as an optional pre-filter).  One thread per (sz+1)^2 block.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_nms_map(const float* __restrict__ src, int M, int N, int sz,
                                                uint8_t* __restrict__ dst) {
  const int nbx = (N + sz) / (sz + 1), nby = (M + sz) / (sz + 1);
  const int b = blockIdx.x * 64 + threadIdx.x;
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

void launch_nms_map(const float* src, int rows, int cols, int sz, uint8_t* dst, hipStream_t s) {
  const int nb = ((cols + sz) / (sz + 1)) * ((rows + sz) / (sz + 1));
  hipMemsetAsync(dst, 0, (size_t)rows * cols, s);
  hipLaunchKernelGGL(k_nms_map, dim3((nb + 63) / 64), dim3(64), 0, s, src, rows, cols, sz, dst);
}
