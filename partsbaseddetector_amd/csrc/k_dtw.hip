// k_dtw.hip — wave-cooperative generalised distance transform: ONE WAVEFRONT PER SCORE LINE,
// bit-identical to the sequential reference (DistanceTransform<T>::computeRow,
// include/DistanceTransform.hpp:151-182).
//
// The reference builds the upper envelope with a stack whose pop decisions compare FLOAT-ROUNDED
// intersection abscissae (`T s = f(...)`, `s <= z[k]`), so the result is not the geometric hull in
// near-degenerate cases and cannot be reproduced by a purely geometric parallel hull.  What can be
// parallelised is (1) guessing and (2) CHECKING the scan's history:
//
//   history   b[q] (q >= 1) = the stack element directly below q at the moment q is pushed;
//             z[q] = S(b[q], q) is then the z stored with q, S = the reference's rounded intersection.
//   theorem   a candidate b with 0 <= b[q] < q equals the sequential history iff
//     (C1) no crossing:  for all q' in (b[q], q):  b[q'] >= b[q]        (b[q] still on the stack at q)
//     (C2) stop test:    b[q] == 0  or  z[q] > z[b[q]]                  (`s <= z[k] && k > 0` fails there)
//     (C3) pop tests:    for every e >= 1 with P(e) = min{q > e : b[q] < e} defined:  S(e, P(e)) <= z[e]
//   (induction over q: the elements popped by q are exactly the live e with b[q] < e < q, i.e. those
//   with P(e) = q; (C3) says the reference pops each of them, (C1)+(C2) that it stops at b[q]).
//   All three use the same S() evaluations the sequential scan performs — about 1.65 per element —
//   but every one of them is independent, so 64 lanes do them at once.
//
//   1. guess     b[q] = argmax_{i<q} s(i,q) — the tangent from q to the envelope of 0..q-1 — ranked in fp32
//                with the cancellation-free key (y_q - y_i)/d - a d, d = q - i.  Lanes run over q, the loop
//                over the distance d ({1/d, -a d} is then wave-uniform): ~N^2/2 fma-class operations.
//   2. verify    z[] (N exact S), sparse table of range-minima of b, P(e) by binary descent, (C1)-(C3).
//   3. repair    if something fails, everything before the first failing q is proven; that q's true
//                b is found by walking the proven live chain with the reference's own pop loop, and
//                step 2 repeats.  (Failures need a float tie or an fp64 near-tie: rare.)  After
//                DTW_MAX_REPAIR rounds lane 0 simply runs the sequential scan for the line.
//   4. envelope  live elements (P undefined) = the final stack; compacted with ballot/popcount.
//   5. read-out  (:172-178) per output by binary search over z, evaluated in fp64 like the reference.
//
// A block = 4 waves = 16 consecutive lines of one map (4 per wave); outputs that must land transposed
// ([q][line]) are staged in LDS and written as 64-byte row segments.
//
// STATUS: selectable (pbd_options.reserved[1] = 2), parity-green, NOT the default.  Measured on MI355X
// (DESIGN.md §5.3): the O(N^2/64) guess costs more wave-time than the lane-per-line scan it replaces
// (k_dt_pass), which stays the product default; this kernel is kept as the verified starting point
// for an O(N log N) guess.
#include "pbd_internal.hpp"

#define DTW_LINES 16        // lines per block
#define DTW_MAX_REPAIR 12

// Lanes exchange data through LDS without a workgroup barrier (one wave owns a line).  The hardware
// keeps a wave's LDS operations in order, but the COMPILER may reorder a load above an earlier store
// whose address it can prove different for the same lane (lane x writes M[x], then reads M[x+h]).
// A wavefront-scope fence emits no instruction and pins program order at every phase boundary.
#define DTW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ unsigned long long pbd_dtw_stats[4];  // [0] lines, [1] repair rounds, [2] sequential fallbacks

__device__ __forceinline__ float dtw_S(double a, double b, double twoa, const double* __restrict__ R,
                                       const float* __restrict__ y, int x0, int x1) {
  // Quadratic::operator()(x0,x1,y0,y1) narrowed to float (DistanceTransform.hpp:98-100,161); same
  // reciprocal-table quotient + exactness guard as k_dt_pass, here with the true division taken inline.
  const int dx = x1 - x0;
  const double dxd = (double)dx;
  const double num = (((double)y[x1] - (double)y[x0]) - b * dxd) + a * (double)(dx * (x1 + x0));
  const double den = twoa * dxd;
  const double r = R[dx];
  const double q0 = num * r;
  const double rem = __builtin_fma(-q0, den, num);
  double q1 = __builtin_fma(rem, r, q0);
  const unsigned long long bits = (unsigned long long)__double_as_longlong(q1);
  const unsigned lo29 = (unsigned)bits & 0x1FFFFFFFu;
  const unsigned ex = (unsigned)(bits >> 52) & 0x7FFu;
  if ((lo29 - 0x0FFFFFFFu) <= 2u || (ex - 897u) > 252u) q1 = num / den;
  return (float)q1;
}

template <typename IDX> struct DtwLds {   // per-wave working set for a line of up to NS elements
  float* y; float* z; float* sy; float* sz; int* misc;
  IDX* b; IDX* P; IDX* sv; IDX* M;  // M: [LOG][NS] range minima of b
};
template <typename IDX> struct IdxTraits;
template <> struct IdxTraits<unsigned char> { static constexpr int NONE = 0xFF; };
template <> struct IdxTraits<unsigned short> { static constexpr int NONE = 0xFFFF; };

// One line, one wavefront.  NCH = number of 64-element chunks (compile time: loops unroll and the
// chunks' LDS reads interleave).  Returns through LDS: stack (sv, sy, sz[0..K+1]), K in misc[1].
template <int NCH, typename IDX>
__device__ __forceinline__ void dtw_line(const DtwLds<IDX>& w, const double* __restrict__ R,
                                         const float2* __restrict__ CT, int len, int LOG, double a, double b, int lane) {
  constexpr int NS = NCH * 64;
  constexpr int NONE = IdxTraits<IDX>::NONE;
  const double twoa = 2 * a;
  // ---- 1. guess: b[q] = argmax_{i<q} s(i,q), the tangent from q to the envelope of 0..q-1.
  // s(i,q) = [ (y_q - y_i)/d + a(2q - d) - b ] / (2a), d = q - i, 2a < 0: maximise s <=> minimise
  // key(d) = (y_q - y_i) * (1/d) - a*d   (per-lane constants dropped; fp32, no cancellation).
  // CT[d] = {1/d, -a*d} is wave-uniform; ties go to the farther element like `s <= z` pops.
  {
    float yq[NCH], best[NCH];
    int bi[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { yq[c] = w.y[min(c * 64 + lane, len - 1)]; best[c] = INFINITY; bi[c] = 0; }
#pragma unroll 4
    for (int d = 1; d < len; ++d) {
      const float2 ct = CT[d];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c * 64 + 63 >= d) {                              // wave-uniform: chunk c still has lanes with q >= d
          const int i = c * 64 + lane - d;
          const float yi = w.y[max(i, 0)];
          const float key = __builtin_fmaf(yq[c] - yi, ct.x, ct.y);
          const bool take = (i >= 0) && (key <= best[c]);
          best[c] = take ? key : best[c];
          bi[c] = take ? i : bi[c];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = c * 64 + lane;
      w.b[q] = (IDX)((q >= 1 && q < len) ? bi[c] : (q == 0 ? 0 : NONE));   // padding never "smaller"
    }
  }
  DTW_SYNC();

  for (int round = 0;; ++round) {
    // ---- 2a. z[q] = S(b[q], q) ----
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = c * 64 + lane;
      if (q >= 1 && q < len) w.z[q] = dtw_S(a, b, twoa, R, w.y, w.b[q], q);
    }
    if (lane == 0) { w.z[0] = -INFINITY; w.misc[0] = 0x7FFFFFFF; }
    DTW_SYNC();
    // ---- 2b. sparse table of range minima of b: M[j][x] = min b[x .. x+2^j-1] ----
#pragma unroll
    for (int c = 0; c < NCH; ++c) { const int x = c * 64 + lane; w.M[x] = (x >= 1) ? w.b[x] : (IDX)NONE; }  // b[0] is not a history entry
    for (int j = 1; j < LOG; ++j) {
      const int h = 1 << (j - 1);
      DTW_SYNC();
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int x = c * 64 + lane;
        const int m0 = w.M[(j - 1) * NS + x];
        const int m1 = (x + h < NS) ? (int)w.M[(j - 1) * NS + x + h] : NONE;
        w.M[j * NS + x] = (IDX)min(m0, m1);
      }
    }
    DTW_SYNC();
    // ---- 2c. P(e) = min{q > e : b[q] < e} by binary descent; conditions (C1)-(C3) ----
    int fail = 0x7FFFFFFF;
    int xs[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) xs[c] = c * 64 + lane + 1;
    for (int j = LOG - 1; j >= 0; --j) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int e = c * 64 + lane, x = xs[c], nx = x + (1 << j);
        if (nx <= NS && x < NS && (int)w.M[j * NS + x] >= e) xs[c] = nx;   // no b < e in [x, x+2^j): skip it
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = c * 64 + lane;
      if (e >= 1 && e < len) {
        const int Pe = (xs[c] < len) ? xs[c] : NONE;
        w.P[e] = (IDX)Pe;
        if (Pe != NONE) {                                     // (C3) e is popped by Pe
          const float s = dtw_S(a, b, twoa, R, w.y, e, Pe);
          if (!(s <= w.z[e])) fail = min(fail, Pe);
        }
        const int be = w.b[e];                                // (C2) stop test at b[e]
        if (be != 0 && !(w.z[e] > w.z[be])) fail = min(fail, e);
        const int lo = be + 1, n = e - lo;                    // (C1) nothing in (b[e], e) points below b[e]
        if (n >= 1) {
          const int j = 31 - __clz(n);
          const int m0 = w.M[j * NS + lo], m1 = w.M[j * NS + e - (1 << j)];
          if (min(m0, m1) < be) fail = min(fail, e);
        }
      }
    }
    if (fail != 0x7FFFFFFF) atomicMin(&w.misc[0], fail);
    DTW_SYNC();
    const int qf = w.misc[0];
    if (qf == 0x7FFFFFFF) break;                       // certificate holds: b is the reference's history
    if (lane == 0) {
      atomicAdd(&pbd_dtw_stats[1], 1ull);
      if (round >= DTW_MAX_REPAIR) {                   // give up guessing: plain sequential scan (:156-170)
        atomicAdd(&pbd_dtw_stats[2], 1ull);
        int top = 0;                                   // live chain is followed through b[]
        for (int q = 1; q < len; ++q) {
          int e = top;
          float s = dtw_S(a, b, twoa, R, w.y, e, q);
          while (s <= w.z[e] && e != 0) { e = w.b[e]; s = dtw_S(a, b, twoa, R, w.y, e, q); }
          w.b[q] = (IDX)e; w.z[q] = s; top = q;
        }
      } else {                                         // everything before qf is proven: redo qf exactly
        int e = qf - 1;                                // top of the live chain before qf
        float s = dtw_S(a, b, twoa, R, w.y, e, qf);
        while (s <= w.z[e] && e != 0) { e = w.b[e]; s = dtw_S(a, b, twoa, R, w.y, e, qf); }
        w.b[qf] = (IDX)e;
      }
    }
    DTW_SYNC();
  }
  // ---- 4. final stack = element 0 + every e with P(e) undefined, in increasing order ----
  int base = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = c * 64 + lane;
    const bool alive = (e < len) && (e == 0 || (int)w.P[e] == NONE);
    const unsigned long long m = __ballot(alive);
    const int k = base + __popcll(m & ((1ull << lane) - 1ull));
    if (alive) { w.sv[k] = (IDX)e; w.sy[k] = w.y[e]; w.sz[k] = w.z[e]; }
    base += __popcll(m);
  }
  if (lane == 0) { w.sz[base] = INFINITY; w.misc[1] = base - 1; }
  DTW_SYNC();
}

// LDS bytes of one block for lines of NS (multiple of 64) elements
__host__ __device__ inline size_t dtw_wave_bytes(int NS, int LOG, int isz) {
  return (((size_t)NS * 4 * 3 + (size_t)(NS + 2) * 4 + 64 + (size_t)NS * isz * (3 + LOG)) + 15) & ~(size_t)15;
}
__host__ __device__ inline void dtw_geom(int len, int* NS, int* LOG, int* isz) {
  *NS = (len + 63) & ~63;
  int lg = 1; while ((1 << lg) < *NS) ++lg;
  *LOG = lg + 1;
  *isz = (len <= 254) ? 1 : 2;
}
size_t dtw_lds_bytes(int len) {
  int NS, LOG, isz; dtw_geom(len, &NS, &LOG, &isz);
  return 4 * dtw_wave_bytes(NS, LOG, isz) + (size_t)NS * (8 + 8) + (size_t)NS * DTW_LINES * (4 + 2) + 64;
}

template <int NCH, typename IDX>
__device__ __forceinline__ void dtw_block(char* smem, const DtTask& t, const DtGroup& g, const DtMap* __restrict__ maps) {
  constexpr int NS = NCH * 64;
  const int len = g.len, nlines = g.nlines;
  int LOG = 1; while ((1 << LOG) < NS) ++LOG;
  LOG += 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = t.g0 / nlines, l0 = t.g0 - mi * nlines;          // blocks never straddle maps
  const int nl = min(DTW_LINES, nlines - l0);
  const DtMap mp = maps[g.map0 + mi];
  const double a = mp.a, b = mp.b;
  // carve LDS
  double* R = (double*)smem;                                      // [NS] 1/(2a d)  (exact, for S)
  float2* CT = (float2*)(R + NS);                                 // [NS] {1/d, -a d} (fp32, for the guess)
  float* so = (float*)(CT + NS);                                  // [NS][DTW_LINES] staged scores
  unsigned short* sp = (unsigned short*)(so + (size_t)NS * DTW_LINES);  // [NS][DTW_LINES] staged pointers
  char* wp = (char*)(sp + (size_t)NS * DTW_LINES);
  wp = (char*)(((size_t)wp + 15) & ~(size_t)15) + dtw_wave_bytes(NS, LOG, (int)sizeof(IDX)) * wave;
  DtwLds<IDX> w;
  w.y = (float*)wp; wp += (size_t)NS * 4;
  w.z = (float*)wp; wp += (size_t)NS * 4;
  w.sy = (float*)wp; wp += (size_t)NS * 4;
  w.sz = (float*)wp; wp += (size_t)(NS + 2) * 4;
  w.misc = (int*)wp; wp += 64;
  w.b = (IDX*)wp; wp += (size_t)NS * sizeof(IDX);
  w.P = (IDX*)wp; wp += (size_t)NS * sizeof(IDX);
  w.sv = (IDX*)wp; wp += (size_t)NS * sizeof(IDX);
  w.M = (IDX*)wp;
  for (int d = tid; d < NS; d += 256) {       // per-map tables: one IEEE division per distance
    R[d] = 1.0 / ((2 * a) * (double)d);
    CT[d] = make_float2(d ? 1.0f / (float)d : 0.f, (float)(-a * (double)d));
  }
  __syncthreads();

  for (int li = wave * 4; li < wave * 4 + 4 && li < nl; ++li) {
    const int line = l0 + li;
    const float* src = (const float*)mp.src + (size_t)line * len;   // float instantiation only
#pragma unroll
    for (int c = 0; c < NCH; ++c) w.y[c * 64 + lane] = src[min(c * 64 + lane, len - 1)];
    DTW_SYNC();
    dtw_line<NCH, IDX>(w, R, CT, len, LOG, a, b, lane);
    // ---- 5. read-out (:172-178): per output, binary search for its envelope piece ----
    const int K = w.misc[1];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = c * 64 + lane;
      if (q < len) {
        const int p = mp.os + q;
        const float fp = (float)p;
        int lo = 0, hi = K;
        while (lo < hi) {                            // smallest k with !(z[k+1] < p); z strictly increasing
          const int mid = (lo + hi) >> 1;
          if (w.sz[mid + 1] < fp) lo = mid + 1; else hi = mid;
        }
        const int v = w.sv[lo];
        const int d = p - v;
        const float out = (float)(a * (double)(d * d) + b * (double)d + (double)w.sy[lo]);
        so[q * DTW_LINES + li] = out;
        if (mp.ptr_natural) mp.ptr[(size_t)line * len + q] = (int16_t)v;   // x pass: row-major, lanes along q
        else sp[q * DTW_LINES + li] = (unsigned short)v;
      }
    }
    DTW_SYNC();
  }
  if (lane == 0 && wave == 0) atomicAdd(&pbd_dtw_stats[0], (unsigned long long)nl);
  __syncthreads();
  // ---- transposed outputs: element q of line i at q*nlines + i, written as nl-element row segments ----
  for (int i = tid; i < len * DTW_LINES; i += 256) {
    const int q = i / DTW_LINES, li = i - q * DTW_LINES;
    if (li < nl) {
      ((float*)mp.dst)[(size_t)q * nlines + l0 + li] = so[i];
      if (!mp.ptr_natural) mp.ptr[(size_t)q * nlines + l0 + li] = (int16_t)sp[i];
    }
  }
}

__global__ __launch_bounds__(256) void k_dt_wave(const DtTask* __restrict__ tasks, const DtGroup* __restrict__ groups,
                                                 const DtMap* __restrict__ maps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const DtTask t = tasks[blockIdx.x];          // t.g0 = first line of this block counted over the group's maps
  const DtGroup g = groups[t.group];
  const int nch = (g.len + 63) >> 6;
  if (g.len <= 254) {
    switch (nch) {
      case 1: dtw_block<1, unsigned char>(smem, t, g, maps); break;
      case 2: dtw_block<2, unsigned char>(smem, t, g, maps); break;
      case 3: dtw_block<3, unsigned char>(smem, t, g, maps); break;
      default: dtw_block<4, unsigned char>(smem, t, g, maps); break;
    }
  } else {
    switch (nch) {
      case 4: dtw_block<4, unsigned short>(smem, t, g, maps); break;
      case 5: dtw_block<5, unsigned short>(smem, t, g, maps); break;
      case 6: dtw_block<6, unsigned short>(smem, t, g, maps); break;
      case 7: dtw_block<7, unsigned short>(smem, t, g, maps); break;
      default: dtw_block<8, unsigned short>(smem, t, g, maps); break;   // len <= 512
    }
  }
}

void launch_dt_wave(const DtTask* tasks, int ntasks, const DtGroup* groups, const DtMap* maps, size_t lds, hipStream_t s) {
  if (ntasks <= 0) return;
  static LdsOptIn optin;
  optin.ensure((const void*)k_dt_wave, lds);
  hipLaunchKernelGGL(k_dt_wave, dim3(ntasks), dim3(256), lds, s, tasks, groups, maps);
}

void dtw_stats_read(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pbd_dtw_stats), sizeof(unsigned long long) * 4);
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(pbd_dtw_stats), z, sizeof(z)); }
}
