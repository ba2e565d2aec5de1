// Ad-hoc probe (not a test): statistics of the segment-parallel DT (dt_core.hpp on the host) on response lines read from lines.bin
// ([int32 nlines, int32 len, float data[nlines * len]]*): scan steps per segment, stitch iterations per boundary, mean and mean-of-max over groups of 64.
//   g++ -O2 -std=c++17 -ffp-contract=off -I partsbaseddetector_amd/csrc tests/tools/dt_line_stats.cpp -o /tmp/stat;  /tmp/stat [budget KB [lanes per block]]
// (tests/tools_dt_line_stats.py dumps the lines from the oracle's responses and runs both block geometries)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
static long g_iter = 0;
#define DT_COUNT_ITER() (++g_iter)
#include "dt_core.hpp"
int main(int argc, char** argv) {
  const int budget_kb = argc > 1 ? atoi(argv[1]) : 25, NT = argc > 2 ? atoi(argv[2]) : 128;
  FILE* f = fopen("lines.bin", "rb");
  if (!f) { fprintf(stderr, "lines.bin not found (tests/tools_dt_line_stats.py writes it)\n"); return 1; }
  const double a = -0.02, b = 0.003;
  int hdr[2];
  while (fread(hdr, 4, 2, f) == 2) {
    const int nl = hdr[0], len = hdr[1];
    std::vector<float> data((size_t)nl * len);
    if (fread(data.data(), 4, data.size(), f) != data.size()) break;
    // block geometry as the planner's (pbd_api.cpp dt_lpb_for, plain groups): lines per block from the LDS budget, rounded to use all lanes
    const int S = (len + 1) | 1;
    const int hdr_bytes = 300 + NT * 12;          // dt_hdr_bytes without the per-line part
    int lpb = std::min(128, (int)((budget_kb * 1024 - hdr_bytes - S * 8) / (S * 9 + 16)));
    lpb = std::max(4, NT / ((NT + lpb - 1) / lpb));
    const int nsub = NT / lpb;
    const int P = dt_segments(nsub, len);
    std::vector<DtPair<float>> YZ(S + 2);
    std::vector<uint8_t> B(S + 2);
    std::vector<double> R(S + 2);
    for (int dx = 1; dx < len; ++dx) R[dx] = 1.0 / dx;
    const double i2a = 1.0 / (2 * a);
    std::vector<long> stit;   // per boundary
    std::vector<long> scan;   // per segment: steps
    long nelem = 0;
    for (int l = 0; l < nl; ++l) {
      for (int i = 0; i < len; ++i) YZ[i].x = data[(size_t)l * len + i];
      std::vector<int> seg(P + 1);
      for (int p = 0; p <= P; ++p) seg[p] = dt_seg_start(p, P, len);
      for (int p = 0; p < P; ++p) {
        // count scan steps: replicate the loop count = pushes + pops = (n-1) + pops
        dt_seg_scan<false, float, uint8_t>(YZ.data(), B.data(), R.data(), i2a, seg[p], seg[p + 1], a, b);
        int pops = 0;
        for (int e = seg[p]; e < seg[p + 1]; ++e) if ((int)B[e] > e) pops++;
        scan.push_back((seg[p + 1] - seg[p] - 1) + pops);
      }
      nelem += len;
      for (int p = P - 1; p >= 1; --p) {
        int f_, dmin, bs; float zs;
        g_iter = 0;
        dt_stitch1<false, float, uint8_t>(YZ.data(), B.data(), R.data(), i2a, seg[p], seg[p + 1], a, b, f_, dmin, zs, bs);
        stit.push_back(g_iter);
      }
    }
    auto wavestat = [](std::vector<long>& v, double& mean, double& wmax) {
      double s = 0; for (long x : v) s += x; mean = v.empty() ? 0 : s / v.size();
      double ws = 0; int nw = 0;
      for (size_t i = 0; i + 64 <= v.size(); i += 64) { ws += *std::max_element(v.begin() + i, v.begin() + i + 64); nw++; }
      wmax = nw ? ws / nw : mean;
    };
    double sm, sw, tm, tw;
    wavestat(scan, sm, sw); wavestat(stit, tm, tw);
    printf("len %3d lines %5d lpb %3d nsub %2d P %2d seglen %.1f | scan steps/seg mean %.1f wave-max %.1f (x%.2f) | stitch iters mean %.1f wave-max %.1f (x%.2f)\n",
           len, nl, lpb, nsub, P, (double)len / P, sm, sw, sw / sm, tm, tw, tm > 0 ? tw / tm : 0.0);
  }
  return 0;
}
