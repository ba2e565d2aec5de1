// dt_core_test.cpp — host-side check of partsbaseddetector_amd/csrc/dt_core.hpp (the segment-parallel exact
// distance transform used by k_dt_pass) against the oracle's sequential loop (oracle/pbd_oracle_T.inc orc_dt1d =
// DistanceTransform.hpp:151-182) on random and adversarial lines: the SAME source the kernel compiles, with the
// lanes of a line run one after the other.  Build + run: tests/test_host_cpu.py::test_dt_core_host (or by hand:
//   g++ -O2 -std=c++17 -ffp-contract=off -I partsbaseddetector_amd/csrc tests/tools/dt_core_test.cpp -L oracle -lorc
//       -Wl,-rpath,$PWD/oracle -o /tmp/dt_core_test && /tmp/dt_core_test 200000)
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <random>
#include <vector>
#include "dt_core.hpp"

extern "C" void orc_dt1d(const float* src, float* dst, int32_t* ptr, int N, double a, double b, int os);
extern "C" void orc_dt1d_f64(const double* src, double* dst, int32_t* ptr, int N, double a, double b, int os);
static void ref1d(const float* s, float* d, int32_t* p, int n, double a, double b, int os) { orc_dt1d(s, d, p, n, a, b, os); }
static void ref1d(const double* s, double* d, int32_t* p, int n, double a, double b, int os) { orc_dt1d_f64(s, d, p, n, a, b, os); }

struct Stats { long lines = 0, suspect = 0, inconsistent = 0, events = 0, redos = 0, fused = 0; };

// one line exactly as k_dt_pass processes it: `lanes` lanes per line
template <typename T, typename IT, bool FZ>   // FZ: the fused arithmetic of float maps with float-born weights (dt_isect's FUSED, the read-out's fused sum)
static void run_line(const T* src, int len, int lanes, double a, double b, int os, T* dst, int32_t* ptr, Stats& st, int order_mode) {
  const int S = (len + 2) & ~1;
  std::vector<DtPair<T>> YZ(S);
  std::vector<IT> B(S), F(lanes), BELOW(lanes);
  std::vector<T> ZLO(lanes);
  std::vector<double> R(S);
  constexpr bool EX = sizeof(T) == 8;
  for (int i = 0; i < len; ++i) YZ[i].x = src[i];
  if (!EX) for (int dx = 1; dx < len; ++dx) R[dx] = 1.0 / (double)dx;   // the block-wide 1/dx table of k_dt_pass
  const double i2a = 1.0 / (2 * a);                                      // DtMap::r2a (host, IEEE)
  int P = dt_segments(lanes, len);
  std::vector<int> seg(P + 1);
  for (int p = 0; p <= P; ++p) seg[p] = dt_seg_start(p, P, len);
  bool flag = false;
  for (int p = 0; p < P; ++p)
    flag |= dt_seg_scan<EX, FZ, T, IT>(YZ.data(), B.data(), R.data(), i2a, seg[p], seg[p + 1], a, b);
  if (flag) st.suspect++;
  if (!flag && P > 1) {
    // the kernel stitches all boundaries concurrently (speculation); any interleaving must give the same result:
    // here right-to-left (every stitch sees completely unstitched neighbours), left-to-right, or shuffled
    std::vector<IT> DM(lanes), BS(lanes);
    std::vector<T> ZS(lanes);
    std::vector<int> order;
    for (int p = 1; p < P; ++p) order.push_back(p);
    if (order_mode % 3 == 0) std::reverse(order.begin(), order.end());
    else if (order_mode % 3 == 2) for (size_t i = 0; i + 1 < order.size(); i += 2) std::swap(order[i], order[i + 1]);
    bool bad = false;
    for (int p : order) {
      int f, dmin, bs;
      T zs;
      bad |= dt_stitch1<EX, FZ, T, IT>(YZ.data(), B.data(), R.data(), i2a, seg[p], seg[p + 1], a, b, f, dmin, zs, bs);
      F[p] = (IT)f; DM[p] = (IT)dmin; ZS[p] = zs; BS[p] = (IT)bs;
    }
    // the validation rounds of k_dt_pass: every lane p >= 2 judges its own stitch against the F its left neighbour's speculative stitch
    // patched (remembered before any redo) and the F the neighbour has now; the lowest stale boundary is redone by its own lane
    std::vector<int> fspec(P, 0);
    for (int p = 2; p < P; ++p) fspec[p] = (int)F[p - 1];
    F[0] = 0;
    bool any = false;
    for (int round = 0; round <= P; ++round) {
      int lowest = -1;
      for (int p = 2; p < P && lowest < 0; ++p)
        if (dt_stitch_stale((int)DM[p], fspec[p], (int)F[p - 1])) lowest = p;
      if (lowest < 0) break;
      if (round == P) { fprintf(stderr, "validation rounds do not terminate\n"); exit(2); }
      any = true;
      int f, bs = (int)BS[lowest];
      T zs = ZS[lowest];
      bad |= dt_stitch_redo<EX, FZ, T, IT>(YZ.data(), B.data(), R.data(), i2a, seg[lowest], seg[lowest + 1], a, b, (int)F[lowest], f, zs, bs);
      F[lowest] = (IT)f; ZS[lowest] = zs; BS[lowest] = (IT)bs; DM[lowest] = (IT)seg[lowest];
      st.redos++;
    }
    if (any) st.events++;
    if (bad) st.inconsistent++;
    flag |= bad;
  }
  if (flag) {                      // fallback: the whole line sequentially, IEEE divisions
    P = 1;
    seg[1] = len;
    dt_seg_scan<true, false, T, IT>(YZ.data(), B.data(), R.data(), i2a, 0, len, a, b);
  }
  F[0] = 0;
  for (int p = 0; p < P; ++p) { BELOW[p] = B[F[p]]; ZLO[p] = YZ[F[p]].y; }   // one lane per segment in the kernel
  const int nsub = lanes, chunk = (len + nsub - 1) / nsub;
  for (int sub = 0; sub < nsub; ++sub) {        // read-out (:172-178), as in the kernel: descending q
    const int q0 = sub * chunk, q1 = std::min(len, q0 + chunk);
    if (q0 >= q1) continue;
    int osq = os + q1 - 1;
    int e = dt_cover<T, IT>(YZ.data(), B.data(), seg.data(), P, BELOW.data(), ZLO.data(), 1, osq);
    for (int q = q1 - 1; q >= q0; --q, --osq) {
      const T fos = (T)osq;
      while (!(YZ[e].y < fos)) e = (int)B[e];
      const int d = osq - e;
      const double dd = (double)d, ad2 = a * (dd * dd);
      dst[q] = (T)((FZ ? fma(b, dd, ad2) : (ad2 + b * dd)) + (double)YZ[e].x);
      ptr[q] = e;
    }
  }
  st.lines++;
}

static int g_kind = -1;
template <typename T>
static int sweep(long nlines, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::uniform_real_distribution<double> ud(0.0, 1.0);
  Stats st;
  std::vector<T> src(2048), d0(2048), d1(2048);
  std::vector<int32_t> p0(2048), p1(2048);
  static const int lens[] = {1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 40, 59, 64, 79, 80, 100, 118, 119, 158, 159, 200, 254, 255, 300, 478, 700};
  static const int lanesv[] = {1, 2, 3, 4, 5, 8, 10, 11, 16, 21, 32, 48, 64};   // lanes per line (a 256-lane block: up to 64)
  for (long it = 0; it < nlines; ++it) {
    const int len = (rng() % 4 == 0) ? 1 + (int)(rng() % 500) : lens[rng() % (sizeof(lens) / sizeof(int))];
    const int lanes = lanesv[rng() % (sizeof(lanesv) / sizeof(int))];
    const int kind = g_kind >= 0 ? g_kind : (int)(rng() % 8);
    double scale = 1.5;
    for (int i = 0; i < len; ++i) {
      double v;
      switch (kind) {
        case 0: v = nd(rng) * scale; break;
        case 1: v = std::round(nd(rng) * 2); break;                          // ties / plateaus
        case 2: v = std::sin(i / 7.0) + 0.05 * nd(rng); break;               // smooth: deep stacks, long pop runs
        case 3: v = ud(rng) * 2e-3 - 1e-3 + (ud(rng) < 0.03 ? 5.0 : 0.0); break;   // sparse peaks
        case 4: v = 0.25 * std::round(nd(rng) * 4); break;                   // quarter steps
        case 5: v = (i % 2) ? 1.0 : 0.0; break;                              // alternating
        case 6: v = -0.01 * (i - len / 2.0) * (i - len / 2.0) * (ud(rng) < 0.5 ? 1 : 0.5) + 0.01 * nd(rng); break;  // concave: everything survives
        default: v = 0.0; break;                                            // constant
      }
      src[i] = (T)v;
    }
    static const double as[] = {1.0, 0.5, 0.25, 0.05, 0.03125, 0.01, 0.007, 0.003, 0.0005, 0.0001};   // the last two: weak curvature, a peak dominates several segments
    // the model's weights are floats (fused arithmetic allowed for float maps: every second such line runs it); one line in eight gets quadratics
    // that are NOT converted floats (pbd_dt2d's caller may hand in any double): unfused only
    const bool anyd = rng() % 8 == 0;
    const double a_ = rng() % 3 == 0 ? as[rng() % 10] : 0.005 + 0.045 * ud(rng);
    const double b_ = rng() % 3 == 0 ? 0.0 : (ud(rng) * 0.02 - 0.01) * (rng() % 4 == 0 ? 5 : 1);
    const double a = anyd ? -a_ : -(double)(float)a_;
    const double b = anyd ? -b_ : -(double)(float)b_;
    const int os = (int)(rng() % 9) - 4;
    const bool fz = sizeof(T) == 4 && !anyd && (it & 1);
    ref1d(src.data(), d0.data(), p0.data(), len, a, b, os);
    if (fz) {
      if (len + 2 <= 256) run_line<T, uint8_t, sizeof(T) == 4>(src.data(), len, lanes, a, b, os, d1.data(), p1.data(), st, (int)(it % 3));
      else run_line<T, uint16_t, sizeof(T) == 4>(src.data(), len, lanes, a, b, os, d1.data(), p1.data(), st, (int)(it % 3));
      st.fused++;
    } else {
      if (len + 2 <= 256) run_line<T, uint8_t, false>(src.data(), len, lanes, a, b, os, d1.data(), p1.data(), st, (int)(it % 3));
      else run_line<T, uint16_t, false>(src.data(), len, lanes, a, b, os, d1.data(), p1.data(), st, (int)(it % 3));
    }
    for (int i = 0; i < len; ++i)
      if (memcmp(&d0[i], &d1[i], sizeof(T)) || p0[i] != p1[i]) {
        fprintf(stderr, "MISMATCH T%zu fused %d len %d lanes %d kind %d a %g b %g os %d at %d: ref (%g,%d) got (%g,%d)\n", sizeof(T), (int)fz, len, lanes,
                kind, a, b, os, i, (double)d0[i], p0[i], (double)d1[i], p1[i]);
        return 1;
      }
  }
  printf("T=%s: %ld lines bit-identical to the sequential reference (%ld redone for a suspect quotient, %ld for a lost stitch invariant; "
         "%ld lines had a speculative stitch re-done, %ld redos; %ld lines with the fused arithmetic)\n", sizeof(T) == 4 ? "float" : "double", st.lines, st.suspect, st.inconsistent, st.events, st.redos, st.fused);
  return 0;
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 100000;
  const unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 1u;
  if (argc > 3) g_kind = atoi(argv[3]);   // restrict the sweep to one kind of line
  if (sweep<float>(n, seed)) return 1;
  if (sweep<double>(n / 2, seed + 1)) return 1;
  return 0;
}
