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
#include "pbd_internal.hpp"

size_t dt_lds_bytes(int stride, int lpb) { return (size_t)lpb * stride * (4 + 4 + 2); }

__device__ __forceinline__ float dt_isect(double a, double b, int x0, int x1, float y0, float y1) {
  // Quadratic::operator()(x0,x1,y0,y1), DistanceTransform.hpp:98-100, narrowed to T at :161
  return (float)((((double)y1 - (double)y0) - b * (double)(x1 - x0) + a * (double)(x1 * x1 - x0 * x0)) /
                 (2 * a * (double)(x1 - x0)));
}

__global__ __launch_bounds__(64) void k_dt_pass(const DtTask* __restrict__ tasks, const DtGroup* __restrict__ groups,
                                                const DtMap* __restrict__ maps, int S, int lpb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Y = (float*)smem;                 // [lpb][S] line values, then y of stack entries (in place)
  float* Z = Y + lpb * S;                  // [lpb][S] z[k], k = 0..len
  unsigned short* V = (unsigned short*)(Z + lpb * S);  // [lpb][S] v[k]
  const DtTask t = tasks[blockIdx.x];
  const DtGroup g = groups[t.group];
  const int lane = threadIdx.x;
  const int len = g.len;
  const int total = g.nmaps * g.nlines;
  const int nl = min(lpb, total - t.g0);
  // cooperative, coalesced load of the nl lines
  for (int i = 0; i < nl; ++i) {
    const int gi = t.g0 + i;
    const int mi = gi / g.nlines, li = gi - mi * g.nlines;
    const float* s = maps[g.map0 + mi].src + (size_t)li * len;
    for (int q = lane; q < len; q += 64) Y[i * S + q] = s[q];
  }
  __syncthreads();
  if (lane >= nl) return;
  const int gi = t.g0 + lane;
  const int mi = gi / g.nlines, li = gi - mi * g.nlines;
  const DtMap mp = maps[g.map0 + mi];
  const double a = mp.a, b = mp.b;
  float* Yl = Y + lane * S;
  float* Zl = Z + lane * S;
  unsigned short* Vl = V + lane * S;

  // ---- build the upper envelope (:156-170) ----
  int k = 0, vk = 0;
  float yk = Yl[0], zk = -INFINITY;
  Vl[0] = 0;
  Zl[0] = -INFINITY;
  for (int q = 1; q < len; ++q) {
    const float yq = Yl[q];
    float s = dt_isect(a, b, vk, q, yk, yq);
    while (s <= zk && k > 0) {
      k--;
      vk = Vl[k]; yk = Yl[k]; zk = Zl[k];
      s = dt_isect(a, b, vk, q, yk, yq);
    }
    k++;
    Vl[k] = (unsigned short)q; Yl[k] = yq; Zl[k] = s;
    vk = q; yk = yq; zk = s;
  }
  Zl[k + 1] = INFINITY;

  // ---- read out (:172-178), transposed + coalesced across lanes ----
  float* dst = mp.dst + li;
  int16_t* ptr = mp.ptr + li;
  const int nlines = g.nlines;
  int os = mp.os;
  k = 0;
  vk = Vl[0]; yk = Yl[0];
  float zn = Zl[1];
  for (int q = 0; q < len; ++q) {
    const float fos = (float)os;
    while (zn < fos) { k++; zn = Zl[k + 1]; vk = Vl[k]; yk = Yl[k]; }
    const int d = os - vk;
    dst[(size_t)q * nlines] = (float)(a * (double)(d * d) + b * (double)d + (double)yk);
    ptr[(size_t)q * nlines] = (int16_t)vk;
    os++;
  }
}

void launch_dt_pass(const DtTask* tasks, int ntasks, const DtGroup* groups, const DtMap* maps, int stride, int lpb,
                    hipStream_t s) {
  if (ntasks <= 0) return;
  const size_t lds = dt_lds_bytes(stride, lpb);
  static size_t configured = 0;
  if (lds > configured) {
    hipFuncSetAttribute((const void*)k_dt_pass, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = lds;
  }
  hipLaunchKernelGGL(k_dt_pass, dim3(ntasks), dim3(64), lds, s, tasks, groups, maps, stride, lpb);
}

// ---------------------------------------------------------------------------
// reduce over child mixtures, one thread per cell of one (level, child part)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce(const ReduceJob* __restrict__ jobs, int njobs,
                                                const float* __restrict__ biasw, int correct_ptr) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  // find the job containing this cell (jobs sorted by cell0; njobs is small): binary search
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].cell0 <= gid) lo = mid; else hi = mid - 1;
  }
  const ReduceJob& J = jobs[lo];
  const unsigned cell = gid - J.cell0;
  const int H = J.H, W = J.W, K = J.K, L = J.L;
  const unsigned HW = (unsigned)H * W;
  if (cell >= HW) return;
  const int m_ = cell / W, n_ = cell - m_ * W;
  for (int m = 0; m < L; ++m) {
    float v;
    int bi = 0;
    if (K == 1) {  // Math::reduceMax K==1 shortcut: copy (Math.hpp:154-158)
      v = J.sdt[cell] + biasw[J.bias_off[0] + m];
    } else {
      v = -INFINITY;
      for (int mm = 0; mm < K; ++mm) {
        const float wv = J.sdt[(size_t)mm * HW + cell] + biasw[J.bias_off[mm] + m];  // DynamicProgram.cpp:139
        if (wv > v) { bi = mm; v = wv; }                      // strict >: first max wins
      }
    }
    int ix = J.ixT[(size_t)bi * HW + (size_t)n_ * H + m_];
    int iy;
    if (!correct_ptr) {
      iy = J.iy[(size_t)bi * HW + (size_t)m_ * W + ix];       // Iy'(m,n) = Iy(m, Ix(m,n))
    } else {
      iy = J.iy[(size_t)bi * HW + cell];
      ix = J.ixT[(size_t)bi * HW + (size_t)n_ * H + iy];      // true arg-max composition
    }
    J.ox[m][cell] = (int16_t)ix;
    J.oy[m][cell] = (int16_t)iy;
    J.ok[m][cell] = (uint8_t)bi;
    J.par_out[m][cell] = J.par_in[m][cell] + v;               // parent.score += maxv (:156)
  }
}

void launch_reduce(const ReduceJob* jobs, int njobs, unsigned total_cells, const float* biasw, int correct_ptr,
                   hipStream_t s) {
  if (njobs <= 0 || total_cells == 0) return;
  hipLaunchKernelGGL(k_reduce, dim3((total_cells + 255) / 256), dim3(256), 0, s, jobs, njobs, biasw, correct_ptr);
}

// ---------------------------------------------------------------------------
// root: bias + max over root mixtures, threshold, compaction
// ---------------------------------------------------------------------------
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
  float v;
  int bi = 0;
  if (J.K == 1) {
    v = J.score[0][cell] + J.bias;
  } else {
    v = -INFINITY;
    for (int m = 0; m < J.K; ++m) {
      const float wv = J.score[m][cell] + J.bias;  // DynamicProgram.cpp:169
      if (wv > v) { bi = m; v = wv; }
    }
  }
  J.rootv[cell] = v;
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
                 int capacity, hipStream_t s) {
  if (njobs <= 0 || total_cells == 0) return;
  hipLaunchKernelGGL(k_root, dim3((total_cells + 255) / 256), dim3(256), 0, s, jobs, njobs, thresh, count, rec,
                     capacity);
}

// ---------------------------------------------------------------------------
// backtrack: one lane per candidate; record = head | boxes[max_parts][4] | locs[max_parts][3]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_backtrack(const int* __restrict__ count, const CandRec* __restrict__ rec,
                                                  int capacity, const BackLevel* __restrict__ back, int ncomp,
                                                  const int* __restrict__ parent, const int* __restrict__ plane0,
                                                  const int* __restrict__ nparts, int max_parts, int kh,
                                                  char* __restrict__ out, size_t out_stride) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
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
  head->score = B.rootv[(size_t)r.y * B.W + r.x];
  head->component = r.comp;
  head->level = r.level;
  head->nparts = np;
  const float scale = B.scale;
  const int sz = __float2int_rn((float)kh * scale);  // Point(xsize,ysize)*scale, cvRound
  for (int p = 0; p < np; ++p) {
    int x, y, m;
    if (p == 0) {
      x = r.x; y = r.y; m = B.rooti[(size_t)r.y * B.W + r.x];
    } else {
      const int par = parent[r.comp * max_parts + p];
      const int px = locs[par * 3], py = locs[par * 3 + 1], pm = locs[par * 3 + 2];
      const size_t off = (size_t)(plane0[r.comp * max_parts + p] + pm) * HW + (size_t)py * B.W + px;
      x = B.px[off]; y = B.py[off]; m = B.pk[off];
    }
    locs[p * 3] = x; locs[p * 3 + 1] = y; locs[p * 3 + 2] = m;
    const int x1 = __float2int_rn((float)(x - 1) * scale), y1 = __float2int_rn((float)(y - 1) * scale);
    const int x2 = x1 + sz - 1, y2 = y1 + sz - 1;
    boxes[p * 4] = min(x1, x2); boxes[p * 4 + 1] = min(y1, y2);
    boxes[p * 4 + 2] = max(x1, x2) - min(x1, x2); boxes[p * 4 + 3] = max(y1, y2) - min(y1, y2);
  }
  for (int p = np; p < max_parts; ++p) {  // components with fewer parts: zero padding
    for (int k = 0; k < 4; ++k) boxes[p * 4 + k] = 0;
    for (int k = 0; k < 3; ++k) locs[p * 3 + k] = 0;
  }
}

void launch_backtrack(const int* count, const CandRec* rec, int capacity, const BackLevel* back, int ncomp,
                      const int* parent, const int* plane0, const int* nparts, int max_parts, int kh, char* out,
                      size_t out_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_backtrack, dim3((capacity + 63) / 64), dim3(64), 0, s, count, rec, capacity, back, ncomp,
                     parent, plane0, nparts, max_parts, kh, out, out_stride);
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
