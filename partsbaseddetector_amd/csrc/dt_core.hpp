// dt_core.hpp — exact, segment-parallel emulation of DistanceTransform<T>::computeRow
// (reference include/DistanceTransform.hpp:151-182, Quadratic::operator() :89-105).
//
// The reference builds the upper envelope of one score line with a sequential stack algorithm whose
// float-rounded intersections decide the result, so the RESULT must be that of the sequential run.  A line is
// long (level 0 of a 640x480 frame: 158 / 118 elements) and a pass lasts as long as its longest line, so the
// line is cut into P segments that P lanes scan CONCURRENTLY, each with the reference's own arithmetic and
// its own (local) stack, and the local results are then stitched into exactly the state the sequential run
// would have reached.  Shared by k_dp.hip (device) and tests/tools/dt_core_test.cpp (host: the same source run
// against the reference loop on millions of random and adversarial lines).
//
// Representation (per line, in LDS on the device): the stack is a LINKED LIST over the line's elements —
//   YZ[e].x  y of element e (the input line, never modified)
//   YZ[e].y  z of e: the intersection computed when e was pushed (`z[k] = s`, :168)
//   B[e]     the element directly below e on the stack at the time e was pushed (`v[k-1]`)
// An element on the stack never changes its z or its "below", so the stack at any time is the chain
// top -> B[top] -> ... -> bottom, and pushing q writes only q's own slots.  9 bytes per element (float).
//
// 1. dt_seg_scan: lane p runs the reference's loop (:156-170) on elements [s_p, s_{p+1}) with an empty initial
//    stack ("local scan"): local z / B for every element of the segment.
//    While it scans, a popped element's B slot (dead from then on) receives the element that popped it:
//    B[e] > e marks e as popped and names its "popper".
// 2. dt_stitch1 (one lane per segment boundary) replays what the GLOBAL run does when it reaches
//    segment B = [s_p, s_{p+1}) with the stack G left by the segments before it.  Invariant: the part of the
//    global stack made of B's elements is a suffix [F, ...] of the local stack, and every entry above F has the
//    same z and the same "below" globally as locally (its predecessor is the same element); only F's z
//    differs (its predecessor lies in G).  For an element q of B the global run therefore performs exactly
//    the local run's pop tests as long as those stay above F — nothing to redo — and deviates only when the
//    local scan of q reached F ("event"): then F is tested with its GLOBAL z:
//      fail -> q is pushed on F, as in the local run iff that also stopped at F; if the local run popped F the
//              invariant is lost: the line is flagged and redone sequentially (only near-degenerate
//              geometry: exact ties / collinear triples);
//      pass -> F is popped and the pops continue into G with explicit tests; q becomes the new F.
//    Events need no search: while F is the top of the local stack the next element is the next event; once an
//    element c sits on F, the next element to reach F is the one that pops c locally — c's popper link.
//    Events are rare (a new "record" element of the segment) and each costs a few intersections.
// 3. Per segment, the element its lowest remaining entry sits on and that entry's z (written by the segment's own
//    lane): enough for a read-out lane to tell which segments still own entries of the final stack and where the
//    chain enters them, i.e. to find the entry covering its last output with a short walk (dt_cover).
// 4. The read-out (:172-178) runs over q in DESCENDING order, sub-ranges in lockstep (coalesced stores), stepping
//    down the chain through the "below" links: k(q) = max{k : z[k] < os + q}, the same entry the reference's
//    ascending `while (z[k+1] < os) k++` reaches because z is strictly increasing along the stack.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define DT_HD __device__ __forceinline__
#define DT_MUL24(a, b) __mul24((a), (b))
#define DT_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)     // any active lane of the wavefront
DT_HD unsigned long long dt_bits(double d) { return (unsigned long long)__double_as_longlong(d); }
#else
#define DT_HD static inline
#define DT_MUL24(a, b) ((a) * (b))
#define DT_ANY(c) (c)
DT_HD unsigned long long dt_bits(double d) { unsigned long long u; memcpy(&u, &d, 8); return u; }
#endif

#ifndef DT_COUNT_ITER
#define DT_COUNT_ITER() ((void)0)   // host-side statistics hook (tests/tools)
#endif
template <typename T> struct alignas(2 * sizeof(T)) DtPair { T x, y; };
// Sticky per-lane "suspect quotient" state: an unsigned MINIMUM accumulated once per intersection (zero = flagged): the two
// tests of dt_isect are arithmetic zero-tests folded into one three-operand minimum per step — no compares, no mask
// registers (a wave-wide lane mask OR-ed per step would have to be copied to vector registers on every iteration of a loop
// the lanes leave at different times).
typedef unsigned DT_SUSPECT_T;
#define DT_SUSPECT_INIT 0xFFFFFFFFu
#define DT_SUSPECT_MINE(acc) ((acc) == 0u)
#ifdef __HIPCC__
DT_HD unsigned dt_min3u(unsigned a, unsigned b, unsigned c) { return min(a, min(b, c)); }
DT_HD unsigned dt_fbits(float f) { return __float_as_uint(f); }
#else
DT_HD unsigned dt_min3u(unsigned a, unsigned b, unsigned c) { unsigned m = a < b ? a : b; return m < c ? m : c; }
DT_HD unsigned dt_fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
#endif

// segment p of P over a line of `len` elements: [dt_seg_start(p), dt_seg_start(p + 1)).  p * len < 2^21 (p <= P <= 64: 256-lane
// blocks at 4 lines per block; len < 2^15): a 32-bit division; the kernel evaluates it once per block into a table
// (seg[0..P], P + 1 <= 65 <= DT_SEGS - 2 entries) that the routines below take.
DT_HD int dt_seg_start(int p, int P, int len) { return (int)((unsigned)(p * len) / (unsigned)P); }
// segments actually used for a line: at least 8 elements each (short lines gain nothing from stitching).  The ONE definition:
// the planner (pbd_api.cpp: dt_group) bakes its value into every task, the host test calls it directly.
#ifdef __HIPCC__
__host__ __device__ inline int dt_segments(int lanes_per_line, int len) {
#else
static inline int dt_segments(int lanes_per_line, int len) {
#endif
  int P = len / 8;
  if (P > lanes_per_line) P = lanes_per_line;
  return P < 1 ? 1 : P;
}

// Intersection of the parabolas rooted at x0 = vk < x1 = q (Quadratic::operator()(x0,x1,y0,y1), :98-100),
// narrowed to T like `T s = f(...)` at :161:
//   num = ((y1 - y0) - b*(x1-x0)) + a*(x1^2 - x0^2)      (same fp64 operations, same order)
//   s   = (T)(num / den),  den = (2a)*(x1-x0)
// EXACT = false (float only): the fp64 division is replaced by TWO multiplications,
//   q1 = RN(RN(num * i2a) * rdx),  i2a = RN(1/(2a)) (per map, an IEEE division on the host),  rdx = RN(1/dx) (a table shared by
// every line of a block, whatever its map).  den = (2a)*dx is exact (a comes from a float, dx < 2^15), so
// q1 = (num/den)(1+e1)(1+e2)(1+e3)(1+e4), |ei| <= 2^-53 (the roundings of i2a, of rdx and of the two products; no
// intermediate under- or overflows unless the result itself leaves the float range): |q1 - num/den| < 4.01 ulp(q1), hence q1
// and RN(num/den) are doubles at most 5 ulp apart, and (float)q1 can differ from (float)RN(num/den) only if a float
// rounding boundary — a double whose low 29 mantissa bits are 0x10000000 — lies between them or on one of them.  The test
// flags every q1 whose low 29 bits lie in [0x0FFFFFF8, 0x10000007] (a boundary within 8 ulp; a power of two next to q1 halves
// the ulp on one side, still inside the window), and every result outside [2^-124, 2^128) in magnitude (the 29-bit spacing
// holds only where (float)q1 is a normal float; the smallest normal binade is left out with a margin, so a q1 just below
// the normal range that rounds up into it is flagged too).  A flagged line is redone with EXACT = true (IEEE division), so
// the result is always bit-identical to the reference's.
// FUSED (round 6; the caller's promise: a and b are doubles converted from FLOATS — the model's weights always are,
// src/DynamicProgram.cpp:125-127 — and the line has at most DT_FUSE_MAXLEN elements): the two products of the numerator are then EXACT in fp64
// (b * dx: 24 + 15 significant bits; a * (x1^2 - x0^2): 24 + 29), so RN(t - RN(b dx)) = RN(t - b dx) = fma(-b, dx, t) and
// RN(t' + RN(a X)) = fma(a, X, t'): the same num, bit for bit, in two instructions fewer and a dependent chain two operations shorter.
#define DT_FUSE_MAXLEN 16384   // x1^2 - x0^2 = dx (x1 + x0) < 2 len^2 <= 2^29; the read-out's d^2 < 2^28 needs |os| + len <= 2^14 as well (the planner checks both)
template <bool EXACT, bool FUSED, typename T>
DT_HD T dt_isect(T yk, int vk, T yq, int q, double a, double b, double twoa, double i2a, double rdx, DT_SUSPECT_T& suspect) {
  const int dx = q - vk;
  const double dxd = (double)dx;
  const double x2 = (double)DT_MUL24(dx, q + vk);                                               // x1^2 - x0^2 < 2^31, operands < 2^16
  const double num = FUSED ? fma(a, x2, fma(-b, dxd, (double)yq - (double)yk))
                           : ((((double)yq - (double)yk) - b * dxd) + a * x2);
  if (EXACT) {
    return (T)(num / (twoa * dxd));
  } else {
    const double q1 = (num * i2a) * rdx;
    const float s = (float)q1;
    // window: ((lo29 ^ 0x10000000) + 8) mod 2^29 < 16  <=>  lo29 in [0x0FFFFFF8, 0x10000007];  range: exponent field of s in
    // {255, 0, 1, 2} (inf / NaN, zero / denormal, |s| < 2^-124: over-covers [2^-125, 2^-124), harmless)  <=>  ((field + 1) & 0xFC) == 0
    const unsigned t = (((unsigned)dt_bits(q1) ^ 0x10000000u) + 8u) & 0x1FFFFFF0u;
    const unsigned r = (dt_fbits(s) + 0x00800000u) & 0x7E000000u;
    suspect = dt_min3u(suspect, t, r);
    return (T)s;
  }
}

// Local scan of elements [s0, s1) (:156-170 with an empty initial stack).  The reference's nested loops
// (for q { while (pop) }) are flattened into a state machine doing exactly one intersection per iteration, so the
// lanes of a wavefront never wait for the slowest lane's pop count and every line sees the reference's sequence
// of intersections / `s <= z[k]` tests in order.  Branch-free body; the state is the stack top (element, y, z), the
// ELEMENT below it, and the element being inserted: what a pop needs of the entry below the top — its (y, z) and its
// own link — is read from LDS at the top of the iteration (the address is known then), together with the reciprocal
// 1/(q - top) and the next line element, and consumed at its end.  The z store goes unconditionally to q's own slot
// (dead when the step pops: q is inserted again).  VALU issue is what bounds the kernel in batches (SQ counters:
// every issue slot of the SIMDs is taken while a pass runs), so the loop carries the fewest selects that do the job.
// RDX[dx] = RN(1 / dx), i2a = RN(1 / (2a)) (EXACT = false only).  Returns the sticky "suspect" flag.
template <bool EXACT, bool FUSED, typename T, typename IT>
DT_HD bool dt_seg_scan(DtPair<T>* __restrict__ YZ, IT* __restrict__ B, const double* __restrict__ RDX, double i2a,
                       int s0, int s1, double a, double b) {
  const double twoa = 2 * a;
  YZ[s0].y = (T)-INFINITY;                  // z[0] = -inf (:158): the bottom of a stack is never popped (`k > 0`, :162)
  B[s0] = (IT)s0;                           // ... and links to itself
  if (s1 - s0 < 2) return false;
  int vk = s0, nv = s0;                     // top of the stack, the entry below it (element indices)
  T zk = (T)-INFINITY;
  T yk = YZ[s0].x;
  int q = s0 + 1;
  T yq = YZ[q].x;
  DT_SUSPECT_T suspect = DT_SUSPECT_INIT;
  while (q < s1) {
    // everything this step may need from LDS (addresses known now, values used after the arithmetic below)
    const DtPair<T> below = YZ[nv];
    const int below_link = (int)B[nv];
    const double rdx = EXACT ? 0.0 : RDX[q - vk];
    const T ynext = YZ[q + 1].x;            // q + 1 == s1 <= len: the slot exists (the stride is >= len + 1) and the value is never used
    const T s = dt_isect<EXACT, FUSED, T>(yk, vk, yq, q, a, b, twoa, i2a, rdx, suspect);
    // :162.  EXACT = false: the bottom's z is -inf, so only s = -inf could pop it — an out-of-range quotient, which
    // dt_isect flags (the line is redone with EXACT = true): no `k > 0` test on this path
    const bool pop = EXACT ? ((s <= zk) && (vk != s0)) : (s <= zk);
    // push: B[q] = top (:166-169).  pop: the popped top's slot is dead from now on and records its popper.
    B[pop ? vk : q] = (IT)(pop ? q : vk);
    YZ[q].y = s;                            // dead if this step pops
    const int vk_o = vk;
    vk = pop ? nv : q;
    nv = pop ? below_link : vk_o;
    yk = pop ? below.x : yq;
    zk = pop ? below.y : s;
    yq = pop ? yq : ynext;
    q += pop ? 0 : 1;
  }
  return DT_SUSPECT_MINE(suspect);
}

// Stitch ONE boundary: replay what the global run does when it reaches segment [s0, s1) with the stack left by
// the elements before s0 (see the header comment).  A flattened state machine doing one intersection per
// iteration; everything an iteration needs from LDS is read in one batch at its top (the entry under test, its
// link, the reciprocal, the links of q and F), so an iteration is one LDS round trip + one intersection.
//
// The P-1 boundaries of a line are stitched CONCURRENTLY, one lane each, i.e. before the segments to the left
// have been stitched: speculation.  What a stitch reads from the left is correct as long as it stays strictly
// above the element F that the left neighbour's own stitch finally leaves as its lowest survivor — entries above
// F have their final z and links from the local scan — so the stitch reports `dmin`, the lowest element below
// s0 it tested, and the validation rounds redo, in order, the few stitches that went
// deeper (rounds: dt_stitch_stale / dt_stitch_redo).  Outputs: f = lowest element of the segment left on the stack, patched to its global z / link (the
// local values are returned in zsave / bsave so that a redo can restore them).  Returns true if the invariant
// was lost or a quotient was suspect (the caller redoes the whole line sequentially).
template <bool EXACT, bool FUSED, typename T, typename IT>
DT_HD bool dt_stitch1(DtPair<T>* __restrict__ YZ, IT* __restrict__ B, const double* __restrict__ RDX, double i2a,
                      int s0, int s1, double a, double b, int& f_out, int& dmin_out, T& zsave, int& bsave) {
  const double twoa = 2 * a;
  DT_SUSPECT_T suspect = DT_SUSPECT_INIT;
  bool bad = false;
  int q = s0;                                // the global run reaches the segment's first element: top of the stack = s0 - 1
  int e = s0 - 1;
  bool testf = false;                        // false: popping below the segment for q; true: e == F, q is the next element whose local scan reached F
  int f = s0, fb = 0, dmin = s0, bf = 0;     // bf: F's LOCAL link / popper (B[f] as the local scan left it: nothing rewrites it before the patch below)
  T zf = (T)0;
  // One step per iteration, branch-free but for the skip of the intersection: the lanes of a wavefront are different
  // stitches in different states, and a divergent branch costs every lane both sides.
  bool done;
  do {
    DT_COUNT_ITER();
    const DtPair<T> ez = YZ[e];
    const DtPair<T> qz = YZ[q];
    const int eb = (int)B[e], bq = (int)B[q];
    // An event whose local scan STOPPED at F (q sits on F locally: F's popper link does not name q) needs no
    // arithmetic: the intersection the global run tests, s(F, q), is the z the local scan stored for q.  Most
    // steps of a stitch are of this kind, and the lanes of a wavefront reach them together (each stitch starts
    // with a few real intersections for the segment's first element): the intersection code is skipped whenever
    // no lane of the wavefront needs it.
    const bool cheap = testf && bf != q;
    T s = qz.y;
    const double rdx = EXACT ? 0.0 : RDX[q - e];
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    // the intersection's operands are fetched with the rest of the iteration's LDS reads: hipcc sinks reads that only the branch below uses INTO the
    // branch — a second LDS round trip in every iteration in which any lane of the wavefront needs the arithmetic (almost every one).  The empty asm
    // pins them here (round 6, session 35: dp_min of a frame alone 0.515 -> 0.502 ms, 0.298 -> 0.295 in batches)
    if constexpr (EXACT) asm volatile("" :: "v"(ez.x), "v"(qz.x));
    else asm volatile("" :: "v"(ez.x), "v"(qz.x), "v"(rdx));
#endif
    if (DT_ANY(!cheap)) {
      const T si = dt_isect<EXACT, FUSED, T>(ez.x, e, qz.x, q, a, b, twoa, i2a, rdx, suspect);   // (a lane that does not need it may flag itself: harmless)
      s = cheap ? s : si;
    }
    dmin = (!testf && e < dmin) ? e : dmin;
    const bool pass = (s <= (testf ? zf : ez.y)) && (e != 0);      // :162; only the bottom of the whole stack is protected
    // pass: F popped (continue below it) or one more pop below the segment; q unchanged, explicit tests from now on.
    // fail, !testf: q is pushed on e: the new F, top of the segment's part of the stack; the next element tests it.
    // fail, testf: F survives q — if the local scan popped it (bf == q: not cheap) the invariant is lost —; q's popper is
    //              the next element to reach F (else none).
    const bool newf = !pass && !testf;
    bad = bad || (!pass && testf && !cheap);
    const int e_pass = testf ? fb : eb;
    const int nq = testf ? (bq > q ? bq : s1) : q + 1;
    f = newf ? q : f;
    zf = newf ? s : zf;
    fb = newf ? e : fb;
    bf = newf ? bq : bf;
    done = !pass && nq >= s1;                // the segment's last event has been handled (the updates below are then unused)
    e = pass ? e_pass : f;
    q = pass ? q : nq;
    testf = !pass;
  } while (!done);
  zsave = YZ[f].y;
  bsave = bf;
  YZ[f].y = zf;
  B[f] = (IT)fb;
  f_out = f;
  dmin_out = dmin;
  return bad || DT_SUSPECT_MINE(suspect);
}

// Validation of the speculative stitches, in ROUNDS, every boundary by its own lane (round 6; rounds 2-5: one lane per line walked
// its P boundaries in order while the block waited — 2 us of a 21 us block on average, 10-18 us in the slowest blocks).
// The only elements whose (z, link) ever change after the local scans are the F of each segment: patched by its speculative
// stitch, and — if that stitch is redone — un-patched again and a (possibly different) F patched instead.  A speculative
// stitch p read elements >= DMIN[p] only, so what it read was final iff DMIN[p] lies strictly above BOTH the F its left
// neighbour's speculative stitch patched (F_spec[p-1]: the patch may have been written, by another wavefront, while p was
// reading) and the F the neighbour finally has (F_new[p-1]); the F of segments further left are lower still.  Boundary 1 is
// always valid (segment 0's local scan IS the global run).
//   round: every lane p >= 2 evaluates dt_stitch_stale() with the F its left neighbour has NOW; the LOWEST stale boundary of
//   a line has only valid — by induction final — boundaries to its left, so its own lane redoes it at once (un-patch, stitch
//   again: dt_stitch_redo), after which it is final; boundaries to its right are judged again in the next round (the one next
//   to it against a new F_new).  Lines proceed independently; a block's rounds end when no lane is stale (typically after the
//   first evaluation: one barrier).
DT_HD bool dt_stitch_stale(int dmin_p, int fspec_prev, int fnew_prev) {
  return dmin_p <= (fspec_prev > fnew_prev ? fspec_prev : fnew_prev);
}
// the redo of boundary [s0, s1) with everything to its left final: fo / zsave / bsave = the speculative stitch's F and its local
// (z, link); returns dt_stitch1's flag, the final F in f_out (patched; its local values in zsave / bsave again)
template <bool EXACT, bool FUSED, typename T, typename IT>
DT_HD bool dt_stitch_redo(DtPair<T>* __restrict__ YZ, IT* __restrict__ B, const double* __restrict__ RDX, double i2a, int s0, int s1,
                          double a, double b, int fo, int& f_out, T& zsave, int& bsave) {
  YZ[fo].y = zsave;                          // undo the speculative patch, then stitch again
  B[fo] = (IT)bsave;
  int dmin;
  return dt_stitch1<EXACT, FUSED, T, IT>(YZ, B, RDX, i2a, s0, s1, a, b, f_out, dmin, zsave, bsave);
}

// After the stitches every segment p has F[p], its lowest element left on the stack when the run finished the
// segment, patched to its global z and link: ZLO[p] = z of F[p], BELOW[p] = the element F[p] sits on (one lane
// per segment writes them, no order).  Later segments may have popped all of segment p; what is then still true:
//   - segment p owns entries of the FINAL stack iff no later segment sits below its start:
//     m_p = min(len - 1, BELOW[p'] : p' > p) >= seg[p]     (pops are permanent and only ever reach further left);
//   - m_p is then the topmost of those entries (the element the next surviving segment above sits on).
// Read-out lane: the stack entry covering output position `osq` (= os + q, compared like `z[k+1] < os`, :174) is
// the topmost entry whose z is < (T)osq: walk the segments top-down carrying m, stop at the first surviving segment
// whose lowest entry qualifies (segment 0 always does: the bottom entry has z = -inf), then step down inside it.
template <typename T, typename IT>
DT_HD int dt_cover(const DtPair<T>* __restrict__ YZ, const IT* __restrict__ B, const int* __restrict__ seg, int P,
                   const IT* __restrict__ BELOW, const T* __restrict__ ZLO, int tstride, int osq) {
  const T fos = (T)osq;
  int m = seg[P] - 1;                        // the last element of the line is the top of the stack
  for (int p = P - 1; p > 0; --p) {
    if (m >= seg[p] && ZLO[p * tstride] < fos) break;
    const int bl = (int)BELOW[p * tstride];
    m = bl < m ? bl : m;
  }
  int e = m;
  while (!(YZ[e].y < fos)) e = (int)B[e];
  return e;
}
