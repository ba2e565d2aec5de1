// Ad-hoc probe (not a test, not part of the product): the filter bank's fp32 products on the bf16 matrix units through EXACT
// three-way bf16 splits (tests/tools_split_products_study.py).  x = h + m + l (three bfloat16, 8 significant
// bits each = the 24 bits of an fp32 significand), both operands; the six partial products above 2^-24 relative
// (hh, hm, mh, hl, lh, mm) go through v_mfma_f32_16x16x32_bf16 into fp32 accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o conv_split_probe conv_split_probe.hip && ./conv_split_probe [cells_w cells_h]
// Prints the maximum error against an fp64 reference on sampled cells and the time / fp32-equivalent TFLOP/s, for NSPLIT = 3
// (6 products) and NSPLIT = 2 (3 products: hh, hl, lh).  The reference figure to beat: k_conv_mfma16 (fp32 MFMA) at 114 TF.
// Implicit GEMM like k_conv_mfma16: M = cells (a workgroup = a 16 x 8 cell tile = 8 M-tiles, 4 wavefronts x 2 M-tiles), N = two
// 16-filter n-tiles per workgroup, K = 25 taps x 32 channels = one 16x16x32 MFMA per (M-tile, n-tile, tap, product).
// Features pre-split in HBM as [cell][split][32 channels] bf16 (192 B per cell; a HOG kernel would write them so), staged into LDS
// with a 208-byte cell stride (16 cells x 16-byte reads hit 64 different banks); filters pre-split as
// [tap][split][k-group 0..3][filter][8] bf16 (one 16-byte load per lane, tap, split and n-tile; L2-resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: an array of them was kept in scratch memory)

constexpr int KH = 5, KW = 5, NTAP = KH * KW, CH = 32;
constexpr int TWO = 16, THO = 8;                       // output tile (cells)
constexpr int TW = TWO + KW - 1, TH = THO + KH - 1;    // halo tile
constexpr int CSTR = 208;                              // LDS bytes per cell (3 x 64 + 16 of padding)

__device__ int g_probe_nostore = 0;   // diagnostics: 1 = results are stored only where they are NaN (never): what do the stores cost?
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// NTW: 16-filter n-tiles per workgroup; SWAP: filters as the A operand, cells as B — D[filter][cell]: the 16 lanes of a k-group then
// hold 16 consecutive cells of ONE response plane per accumulator register (64-byte store segments without an LDS transpose)
template <int NSPLIT, int NTW, bool SWAP>
__global__ __launch_bounds__(256, NTW > 2 ? 2 : 3) void k_conv_split(const u16* __restrict__ feat, const u16* __restrict__ filt, float* __restrict__ resp,
                                                       int W, int H, int nf, int nfpad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + TWO - 1) / TWO;
  const int tile = blockIdx.x, ty0 = (tile / tiles_x) * THO, tx0 = (tile % tiles_x) * TWO;
  const int nbase = blockIdx.y * 16 * NTW;
  // ---- stage the halo tile: NSPLIT x 64 B per cell, 16-byte pieces ----
  constexpr int PPC = NSPLIT * 4;                      // 16-byte pieces per cell
  {  // all pieces of the thread in flight at once (a load -> store loop is one memory round trip per piece: 11 in a row took as long
     // as the workgroup's MFMAs); addresses clamped, the zero border selected afterwards
    constexpr int NPC = TH * TW * PPC, NB = (NPC + 255) / 256;
    u32x4 v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = min(tid + j * 256, NPC - 1);
      const int cell = i / PPC, piece = i - cell * PPC;
      const int cy = cell / TW, cx = cell - cy * TW;
      const int y = min(max(ty0 + cy - KH / 2, 0), H - 1), x = min(max(tx0 + cx - KW / 2, 0), W - 1);
      v[j] = *(const u32x4*)(feat + ((size_t)(y * W + x) * 3 * CH + piece * 8));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = tid + j * 256;
      if (i < NPC) {
        const int cell = i / PPC, piece = i - cell * PPC;
        const int cy = cell / TW, cx = cell - cy * TW;
        const int y = ty0 + cy - KH / 2, x = tx0 + cx - KW / 2;
        const bool inside = y >= 0 && y < H && x >= 0 && x < W;
        *(u32x4*)(smem + cell * CSTR + piece * 16) = inside ? v[j] : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  __syncthreads();
  const int ai = lane & 15, g = lane >> 4;             // A row / B column of the lane, k-group
  f32x4 acc[NTW][2];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[nt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  // M-tile mt = 2 * wave + m covers output row mt of the tile (16 cells of one row)
  const u16* bl = filt + ((size_t)g * nfpad + nbase + ai) * 8;
  // B one tap ahead in registers (the filters come from L2: a global round trip per tap would otherwise stand in front of 24 MFMAs)
  auto load_b = [&](bf16x8 (&b)[NTW][NSPLIT], int tap) {
    const int t = min(tap, NTAP - 1);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int s = 0; s < NSPLIT; ++s)
        b[nt][s] = *(const bf16x8*)(bl + ((size_t)(t * 3 + s) * 4 * nfpad + 16 * nt) * 8);
  };
  auto mma_tap = [&](const bf16x8 (&b)[NTW][NSPLIT], int tap) {
    const int ti = tap / KW, tj = tap - ti * KW;
    bf16x8 a[2][NSPLIT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int s = 0; s < NSPLIT; ++s)
        a[m][s] = *(const bf16x8*)(smem + ((2 * wave + m + ti) * TW + ai + tj) * CSTR + s * 64 + g * 16);
    // products outermost: consecutive MFMAs go to DIFFERENT accumulators (six in a row on one accumulator wait for each other);
    // per accumulator the order is still small terms first
    auto sweep = [&](int sa, int sb) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          acc[nt][m] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt][sb], a[m][sa], acc[nt][m], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][sa], b[nt][sb], acc[nt][m], 0, 0, 0);
    };
    if constexpr (NSPLIT == 3) { sweep(1, 1); sweep(0, 2); sweep(2, 0); }
    sweep(0, 1); sweep(1, 0); sweep(0, 0);
  };
  bf16x8 b0[NTW][NSPLIT], b1[NTW][NSPLIT];
  load_b(b0, 0);
#pragma unroll 1
  for (int tap = 0; tap < NTAP; tap += 2) {
    load_b(b1, tap + 1);
    mma_tap(b0, tap);
    if (tap + 1 < NTAP) {
      load_b(b0, tap + 2);
      mma_tap(b1, tap + 1);
    }
  }
  // ---- store.  !SWAP: D[i = 4 g + r][j = ai] = (cell 4 g + r of the M-tile's row, filter 16 nt + ai): 4-byte pieces of 16 planes per
  // instruction.  SWAP: D[i][j] = (filter 16 nt + 4 g + r, cell ai): 64 contiguous bytes per k-group and register ----
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y = ty0 + 2 * wave + m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fn = nbase + 16 * nt + (SWAP ? 4 * g + r : ai);
        const int x = tx0 + (SWAP ? ai : 4 * g + r);
        if (fn < nf && y < H && x < W && (!g_probe_nostore || acc[nt][m][r] != acc[nt][m][r])) resp[(size_t)fn * H * W + (size_t)y * W + x] = acc[nt][m][r];
      }
    }
}

// Variant: the filters' tap slab goes through LDS once per workgroup (the four wavefronts of a workgroup use the same filters: loaded
// by every wavefront separately they cost four times the L2 traffic — 18 TB/s in the variant above), double-buffered, one barrier per
// tap; D[filter][cell] as above.  NTW n-tiles per workgroup.
template <int NSPLIT, int NTW>
__global__ __launch_bounds__(256, 2) void k_conv_split_blds(const u16* __restrict__ feat, const u16* __restrict__ filt, float* __restrict__ resp,
                                                            int W, int H, int nf, int nfpad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BSLAB = NSPLIT * 4 * NTW * 16 * 16;       // bytes of one tap's slab: [split][k-group][16 NTW filters][8 bf16]
  char* bbuf = smem + TH * TW * CSTR;                      // two slabs behind the feature tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + TWO - 1) / TWO;
  const int tile = blockIdx.x, ty0 = (tile / tiles_x) * THO, tx0 = (tile % tiles_x) * TWO;
  const int nbase = blockIdx.y * 16 * NTW;
  constexpr int PPC = NSPLIT * 4;
  for (int i = tid; i < TH * TW * PPC; i += 256) {
    const int cell = i / PPC, piece = i - cell * PPC;
    const int cy = cell / TW, cx = cell - cy * TW;
    const int y = ty0 + cy - KH / 2, x = tx0 + cx - KW / 2;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (y >= 0 && y < H && x >= 0 && x < W) v = *(const u32x4*)(feat + ((size_t)(y * W + x) * 3 * CH + piece * 8));
    *(u32x4*)(smem + cell * CSTR + piece * 16) = v;
  }
  // slab pieces of this thread: piece p = (split s, k-group g, filter f of the workgroup's 16 NTW): source row (tap * 3 + s) * 4 + g
  constexpr int NPIECE = NSPLIT * 4 * NTW * 16, PPT = (NPIECE + 255) / 256;
  u32x4 breg[PPT];
#define LOAD_SLAB(tap_) do { _Pragma("unroll") for (int j = 0; j < PPT; ++j) { \
    const int p = min(tid + j * 256, NPIECE - 1); \
    const int sg = p / (NTW * 16), f = p - sg * (NTW * 16), s_ = sg >> 2, g_ = sg & 3; \
    breg[j] = *(const u32x4*)(filt + ((((size_t)(tap_) * 3 + s_) * 4 + g_) * nfpad + nbase + f) * 8); } } while (0)
#define STORE_SLAB(buf_) do { _Pragma("unroll") for (int j = 0; j < PPT; ++j) { \
    const int p = tid + j * 256; \
    if (p < NPIECE) *(u32x4*)(bbuf + (buf_) * BSLAB + p * 16) = breg[j]; } } while (0)
  LOAD_SLAB(0);
  STORE_SLAB(0);
  __syncthreads();
  const int ai = lane & 15, g = lane >> 4;
  f32x4 acc[NTW][2];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[nt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int tap = 0; tap < NTAP; ++tap) {
    if (tap + 1 < NTAP) LOAD_SLAB(tap + 1);
    const int ti = tap / KW, tj = tap - ti * KW;
    const char* bs = bbuf + (tap & 1) * BSLAB;
    bf16x8 a[2][NSPLIT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int s = 0; s < NSPLIT; ++s)
        a[m][s] = *(const bf16x8*)(smem + ((2 * wave + m + ti) * TW + ai + tj) * CSTR + s * 64 + g * 16);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      bf16x8 b[NSPLIT];
#pragma unroll
      for (int s = 0; s < NSPLIT; ++s) b[s] = *(const bf16x8*)(bs + (((s * 4 + g) * NTW + nt) * 16 + ai) * 16);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        auto mm = [&](const bf16x8& x, const bf16x8& w) { acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc[nt][m], 0, 0, 0); };
        if constexpr (NSPLIT == 3) { mm(a[m][1], b[1]); mm(a[m][0], b[2]); mm(a[m][2], b[0]); }
        mm(a[m][0], b[1]); mm(a[m][1], b[0]); mm(a[m][0], b[0]);
      }
    }
    if (tap + 1 < NTAP) STORE_SLAB((tap + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y = ty0 + 2 * wave + m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fn = nbase + 16 * nt + 4 * g + r, x = tx0 + ai;
        if (fn < nf && y < H && x < W) resp[(size_t)fn * H * W + (size_t)y * W + x] = acc[nt][m][r];
      }
    }
}

static u16 bf16_rn(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (u16)(u >> 16); }
static float bf16_f(u16 h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(float x, u16 out[3]) { float r = x; for (int s = 0; s < 3; ++s) { out[s] = bf16_rn(r); r -= bf16_f(out[s]); } }

template <typename K>
static void run(const char* name, K kernel, size_t lds, int NTW, const u16* d_feat, const u16* d_filt, float* d_resp, int W, int H, int nf, int nfpad, const std::vector<float>& F,
                const std::vector<float>& Wt, std::vector<float>& out) {
  CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(((W + TWO - 1) / TWO) * ((H + THO - 1) / THO), (nf + 16 * NTW - 1) / (16 * NTW));
  CHECK(hipMemset(d_resp, 0xff, (size_t)nf * W * H * 4));       // (NaN pattern: a cell the kernel does not write shows up in the check)
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, grid, dim3(256), lds, 0, d_feat, d_filt, d_resp, W, H, nf, nfpad);
  CHECK(hipDeviceSynchronize());
  const int reps = 5;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, grid, dim3(256), lds, 0, d_feat, d_filt, d_resp, W, H, nf, nfpad);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  CHECK(hipMemcpy(out.data(), d_resp, out.size() * 4, hipMemcpyDeviceToHost));
  // fp64 reference on sampled cells (zero border)
  std::mt19937 rng(7);
  double maxerr = 0, maxref = 0;
  for (int t = 0; t < 600; ++t) {
    const int y = t < 40 ? (t % 2 ? H - 1 - t / 8 : t / 8) : (int)(rng() % H), x = t < 40 ? (t % 3 ? W - 1 - t / 8 : t / 8) : (int)(rng() % W);
    for (int fn = 0; fn < nf; fn += (t % 7) + 1) {
      double s = 0;
      for (int ti = 0; ti < KH; ++ti) for (int tj = 0; tj < KW; ++tj) {
        const int yy = y + ti - KH / 2, xx = x + tj - KW / 2;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        for (int c = 0; c < CH; ++c) s += (double)F[((size_t)yy * W + xx) * CH + c] * (double)Wt[(((size_t)fn * KH + ti) * KW + tj) * CH + c];
      }
      maxerr = std::max(maxerr, std::fabs(s - (double)out[(size_t)fn * H * W + (size_t)y * W + x]));
      maxref = std::max(maxref, std::fabs(s));
    }
  }
  const double flop = 2.0 * W * H * (double)nf * NTAP * CH;
  printf("%-42s %.3f ms  %.1f fp32-equivalent TFLOP/s  max |err| vs fp64 %.3e (|response| max %.3f)\n", name, ms, flop / ms / 1e9, maxerr, maxref);
}

int main(int argc, char** argv) {
  const int W = argc > 2 ? atoi(argv[1]) : 1024, H = argc > 2 ? atoi(argv[2]) : 1024, nf = 156, nfpad = 160;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> ud(0.f, 0.4f);
  std::normal_distribution<float> nd(0.f, 0.05f);
  std::vector<float> F((size_t)W * H * CH), Wt((size_t)nf * NTAP * CH);
  for (auto& v : F) v = ud(rng) < 0.1f ? 0.f : ud(rng);
  for (auto& v : Wt) v = nd(rng);
  std::vector<u16> Fs((size_t)W * H * 3 * CH), Ws((size_t)NTAP * 3 * 4 * nfpad * 8, 0);
  for (size_t cell = 0; cell < (size_t)W * H; ++cell)
    for (int c = 0; c < CH; ++c) { u16 p[3]; split3(F[cell * CH + c], p); for (int s = 0; s < 3; ++s) Fs[(cell * 3 + s) * CH + c] = p[s]; }
  for (int fn = 0; fn < nf; ++fn)
    for (int tap = 0; tap < NTAP; ++tap)
      for (int c = 0; c < CH; ++c) {
        u16 p[3]; split3(Wt[((size_t)fn * NTAP + tap) * CH + c], p);
        for (int s = 0; s < 3; ++s) Ws[((((size_t)tap * 3 + s) * 4 + c / 8) * nfpad + fn) * 8 + c % 8] = p[s];
      }
  u16 *d_feat, *d_filt; float* d_resp;
  CHECK(hipMalloc(&d_feat, Fs.size() * 2)); CHECK(hipMalloc(&d_filt, Ws.size() * 2)); CHECK(hipMalloc(&d_resp, (size_t)nf * W * H * 4));
  CHECK(hipMemcpy(d_feat, Fs.data(), Fs.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_filt, Ws.data(), Ws.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> out((size_t)nf * W * H);
  printf("%d x %d cells, %d filters 5x5x32 (K = 800)\n", W, H, nf);
  const size_t ldsA = (size_t)TH * TW * CSTR;
#define RUN(name, NS, NT, SW) run(name, k_conv_split<NS, NT, SW>, ldsA, NT, d_feat, d_filt, d_resp, W, H, nf, nfpad, F, Wt, out)
#define RUNB(name, NS, NT) run(name, k_conv_split_blds<NS, NT>, ldsA + 2 * (size_t)(NS * 4 * NT * 16 * 16), NT, d_feat, d_filt, d_resp, W, H, nf, nfpad, F, Wt, out)
  RUN("6 products, 2 n-tiles", 3, 2, false);
  RUN("6 products, 5 n-tiles, D[filter][cell]", 3, 5, true);
  RUNB("6 products, 5 n-tiles, B through LDS", 3, 5);
  RUNB("6 products, 10 n-tiles, B through LDS", 3, 10);
  RUN("3 products, 5 n-tiles, D[filter][cell]", 2, 5, true);
  RUNB("3 products, 5 n-tiles, B through LDS", 2, 5);
  RUNB("3 products, 10 n-tiles, B through LDS", 2, 10);
  { const int one = 1; CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_nostore), &one, sizeof(int))); }
  printf("-- diagnostics: the same kernels without their stores (the error column is meaningless) --\n");
  RUN("6 products, 5 n-tiles, NO stores", 3, 5, true);
  RUN("3 products, 5 n-tiles, NO stores", 2, 5, true);
  return 0;
}
