// Ad-hoc probe (not a test): how many validation rounds the speculative stitches of REAL response lines need (dt_core.hpp on the host; lines.bin as
// written by tests/tools_dt_line_stats.py) under two protocols: "lowest first" (k_dt_pass, round 6: the lowest stale boundary of a line per round) and
// "all at once" (every stale boundary of a line redone in the same round, again speculatively).  Per geometry: share of lines / blocks with a stale
// boundary, rounds per affected line and per affected block (a block waits for its slowest line).
//   g++ -O2 -std=c++17 -ffp-contract=off -I partsbaseddetector_amd/csrc tests/tools/dt_rounds_stats.cpp -o /tmp/rounds; /tmp/rounds [budget KB [lanes]]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "dt_core.hpp"
typedef uint8_t IT;
struct Line { std::vector<DtPair<float>> YZ; std::vector<IT> B; };
int main(int argc, char** argv) {
  const int budget_kb = argc > 1 ? atoi(argv[1]) : 40, NT = argc > 2 ? atoi(argv[2]) : 256;
  FILE* f = fopen("lines.bin", "rb");
  if (!f) return 1;
  const double a = argc > 3 ? -atof(argv[3]) : -0.02, b = 0.003, i2a = 1.0 / (2 * a);
  int hdr[2];
  while (fread(hdr, 4, 2, f) == 2) {
    const int nl = hdr[0], len = hdr[1];
    std::vector<float> data((size_t)nl * len);
    if (fread(data.data(), 4, data.size(), f) != data.size()) break;
    const int S = (len + 1) | 1;
    const int hdr_bytes = 300 + NT * 12;
    int lpb = std::min(128, (int)((budget_kb * 1024 - hdr_bytes - S * 8) / (S * 9 + 16)));
    lpb = std::max(4, NT / ((NT + lpb - 1) / lpb));
    const int nsub = NT / lpb, P = dt_segments(nsub, len);
    std::vector<double> R(S + 2);
    for (int dx = 1; dx < len; ++dx) R[dx] = 1.0 / dx;
    std::vector<int> seg(P + 1);
    for (int p = 0; p <= P; ++p) seg[p] = dt_seg_start(p, P, len);
    long stale_lines = 0, rounds_low = 0, rounds_all = 0, nbound = 0, nstale0 = 0;
    std::vector<int> rl((size_t)nl, 0), ra((size_t)nl, 0);
    for (int l = 0; l < nl; ++l) {
      for (int proto = 0; proto < 2; ++proto) {
        std::vector<DtPair<float>> YZ(S + 2);
        std::vector<IT> B(S + 2);
        for (int i = 0; i < len; ++i) YZ[i].x = data[(size_t)l * len + i];
        bool flag = false;
        for (int p = 0; p < P; ++p) flag |= dt_seg_scan<false, float, IT>(YZ.data(), B.data(), R.data(), i2a, seg[p], seg[p + 1], a, b);
        if (flag || P < 3) continue;
        std::vector<int> F(P, 0), DM(P, 0), BS(P, 0), FS(P, 0);
        std::vector<float> ZS(P, 0);
        bool bad = false;
        for (int p = P - 1; p >= 1; --p) { int f_, dm, bs; float zs; bad |= dt_stitch1<false, float, IT>(YZ.data(), B.data(), R.data(), i2a, seg[p], seg[p + 1], a, b, f_, dm, zs, bs); F[p] = f_; DM[p] = dm; ZS[p] = zs; BS[p] = bs; }
        if (bad) continue;
        for (int p = 2; p < P; ++p) FS[p] = F[p - 1];
        int rounds = 0;
        for (;; ++rounds) {
          std::vector<int> st;
          for (int p = 2; p < P; ++p) if (dt_stitch_stale(DM[p], FS[p], F[p - 1])) st.push_back(p);
          if (st.empty()) break;
          if (rounds == 0 && proto == 0) { nstale0 += (long)st.size(); }
          if (proto == 0) st.resize(1);
          // "all at once": descending order = every redo sees its left neighbours' state of the round's start (the worst interleaving)
          std::vector<int> Fstart(F);
          for (int i = (int)st.size() - 1; i >= 0; --i) {
            const int p = st[i];
            int f_, dm, bs = BS[p]; float zs = ZS[p];
            YZ[F[p]].y = zs; B[F[p]] = (IT)bs;
            dt_stitch1<false, float, IT>(YZ.data(), B.data(), R.data(), i2a, seg[p], seg[p + 1], a, b, f_, dm, zs, bs);
            F[p] = f_; ZS[p] = zs; BS[p] = bs; DM[p] = dm; FS[p] = Fstart[p - 1];
          }
          DM[st[0]] = seg[st[0]];          // the lowest stale boundary had only final boundaries to its left: final, never stale again
          if (rounds > 4 * P) { fprintf(stderr, "no convergence\n"); return 2; }
        }
        (proto ? ra : rl)[(size_t)l] = rounds;
        if (proto == 0) { nbound += P - 2; if (rounds) { stale_lines++; rounds_low += rounds; } } else rounds_all += rounds;
      }
    }
    long nblk = 0, blk_stale = 0, blk_rl = 0, blk_ra = 0;
    for (int l0 = 0; l0 < nl; l0 += lpb) {
      int ml = 0, ma = 0;
      for (int l = l0; l < std::min(nl, l0 + lpb); ++l) { ml = std::max(ml, rl[(size_t)l]); ma = std::max(ma, ra[(size_t)l]); }
      nblk++; if (ml) { blk_stale++; blk_rl += ml; blk_ra += ma; }
    }
    printf("len %3d lines %5d lpb %3d P %2d | stale boundaries %.3f %% | lines with one %.2f %%: rounds per such line %.2f (lowest first) / %.2f (all at once) | blocks with one %.1f %%: rounds per such block %.2f / %.2f\n",
           len, nl, lpb, P, 100.0 * nstale0 / std::max(1L, nbound), 100.0 * stale_lines / nl, stale_lines ? (double)rounds_low / stale_lines : 0.0,
           stale_lines ? (double)rounds_all / stale_lines : 0.0, 100.0 * blk_stale / nblk, blk_stale ? (double)blk_rl / blk_stale : 0.0, blk_stale ? (double)blk_ra / blk_stale : 0.0);
  }
  return 0;
}
