// pbd_internal.hpp — shared declarations of libpbd_hip.so (gfx950 only).
//
// Data layout in HBM for one frame (all buffers owned by the handle, sized on
// the first frame of a given geometry and reused):
//   img      u8   source image, tightly packed w*cn
//   pyr      u8   level images back to back (level l at img_off[l])
//   feat     f32  HOG: level l at feat_off[l], [ch][cw][32]      (cell-major)
//   resp     f32  pdf: level l at resp_off[l], [nfilters][ch][cw] (plane-major)
//   acc      f32  accumulated part scores: level l at acc_off[l], [nslots][ch][cw]
//   ptrk     u8   DP best child mixture Ik: level l at cell_off[l]*nplanes, [nplanes][ch][cw]
//                 (Ix / Iy are composed at back-tracking time from the DT pointer planes dt_ixT / dt_iy)
//   rootv/i  f32/i32  level l at root_off[l], [ncomp][ch][cw]
//   dt_*          per-round scratch of the distance transform
//   cand          device candidate list (count + records)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/pbd_c.h"

#define PBD_MAX_LEVELS 128
#define PBD_FLEN 32
#define PBD_NORIENT 18
#define PBD_MAX_MIX 16

struct Level {
  int iw, ih;      // level image size
  int bw, bh;      // HOG blocks
  int cw, ch;      // cells (feature map / response size)
  float scale;     // IFeatures::scales()[l]
  size_t img_off;  // bytes into pyr
  size_t cell_off; // prefix sum of cells (over all levels)
  int active;      // within [level_begin, level_end)
};

// ---- kernel work tables -----------------------------------------------------
// one image of the pyramid: level image (frame f, level l) from the frame itself (cv::resize, first octave) or from
// level l - interval (cv::pyrDown); tables in device memory, one launch covers every job of a stage
struct PyrJob { unsigned long long soff, doff; int sw, sh, dw, dh; };
struct HogTile { int level, cy0, cx0, pad; };
struct LevelDev {    // per level, device copy
  int iw, ih, bw, bh, cw, ch;
  unsigned long long img_off, cell_off;
};
struct ConvTile { int level, y0, x0, pad; };

// Score data is T = float or double (the handle's instantiation, pbd_options.scalar_type); the work
// tables carry untyped pointers and the kernels are instantiated for both.
struct DtMap {       // one 1-D pass over one score map
  const void* src;   // T: lines contiguous: line i at src + i*len
  void* dst;         // T: transposed out: element q of line i at dst + q*nlines + i
  int16_t* ptr;      // same layout as dst
  double a, b;       // Quadratic(a, b)
  double r2a;        // RN(1 / (2a)), IEEE division on the host (dt_core.hpp: the reciprocal of an intersection's denominator)
  int os, ptr_natural;  // ptr_natural: write ptr row-major [line][q] instead of transposed
};
// One group = the maps of one launch that share a geometry: nmaps maps of nlines lines of len elements.
//   plain:  line gi of the group = line gi % nlines of map gi / nlines (map-major); a block = lpb consecutive lines.
//   fold:   the group is ONE part at one level (nmaps = its K mixtures, the lines of a row are its K mixtures): a
//           block = nrows consecutive rows x K mixtures, and its loader builds the lines on the fly from the part's raw
//           responses and its children's distance-transformed scores (FoldJob).
// The block's index arithmetic divides by wave-uniform numbers (lines per block, lines per map, segments per line): the plan
// supplies them as multiply-high constants — a division by a run-time value is ~20 vector instructions per lane, and vector
// instruction issue is what bounds k_dt_pass.  magic_d = ceil(2^32 / d): x / d == umulhi(x, magic_d) exactly for x * d < 2^32.
struct DtGroup {
  int map0, nmaps, nlines, len, stride, lpb, fold;   // stride: LDS elements per line (odd); lpb: lines per block; fold: FoldJob index or -1
  int nsub;                  // lanes per line = block lanes / lpb
  int P;                     // segments per line = dt_segments(nsub, len)
  int chunk;                 // read-out: outputs per lane = ceil(len / nsub)
  unsigned magic_lpb;        // lane / lpb              (lane < 2^8)
  unsigned magic_nlines;     // (l0 + lane) / nlines    (numerator < nlines + 2^8, nlines < 2^15)
  unsigned magic_P;          // (p * len) / P           (p * len < 2^21, P <= 64)
  int fused;                 // bit 0 — float maps only: every map of the group has weights that are converted floats, len and len + |os| <= DT_FUSE_MAXLEN: the
                             // intersection's and the read-out's products are exact and fuse into their additions (dt_core.hpp: dt_isect);
                             // bit 1 — the group's maps write their pointers in natural layout (DtMap::ptr_natural of every map of the group: x passes): the
                             // block reads it HERE — from the lane's map descriptor hipcc evaluated it right behind the descriptor's load, a full memory round
                             // trip in front of the loader of every block
};
#define DT_G_FUSED 1
#define DT_G_NATURAL 2
struct DtTask { int g0, nl, m0, l0; DtGroup g;     // g0: first line (plain) / first row (fold); nl: lines of this block; plain: g0 = m0 * nlines + l0
                                                    // (first map of the block, first line inside it); the group travels with the task
  const void* src0; };                              // plain: the block's first line when its nl lines are CONTIGUOUS in memory (consecutive maps of a group
                                                    // back to back — the y pass's input always, plan_frame), else null: the loader then needs no map descriptor
static inline unsigned dt_magic(unsigned d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; }   // d == 1: the quotient is the numerator itself (callers test)
#ifndef PBD_ARGMIN_ZERO_COPY
#define PBD_ARGMIN_ZERO_COPY 1   // k_backtrack writes candidates straight to the pinned host buffers (handles outside RCCL groups)
#endif
#ifndef PBD_DT_PRIO
#define PBD_DT_PRIO 1            // s_setprio of k_dt_pass's wavefronts (k_dp.hip)
#endif
#ifndef PBD_DT_TASK_PREFETCH
#define PBD_DT_TASK_PREFETCH 256 // k_dt_pass touches the task descriptor this many blocks ahead (k_dp.hip; 0: off)
#endif
#ifndef PBD_DT_NT_DEFAULT
#define PBD_DT_NT_DEFAULT 128   // lanes of a k_dt_pass block
#endif

#define PBD_MAX_CH 8   // children of one parent folded into one reduce job
struct ReduceChild {     // one child part's distance-transformed mixtures
  const void* sdt;       // T [K][H][W] distance-transformed child scores
  uint8_t* ok;           // output: best child mixture per parent mixture, [L][H][W].  The x / y pointers of the
                         // winning mixture are NOT materialised: the DT pointer planes stay in HBM for the frame and
                         // the few back-tracked candidates compose them on the fly (k_backtrack)
  int K, pad;
  int bias_off[PBD_MAX_MIX];  // biasw index of bias(mm)[0] for each child mixture mm
};
// fold mode: the children of one (level, part), descending child index (src/DynamicProgram.cpp:95); read by the loader
// of the part's x pass (k_dt_pass<T, true>) and, for a root, by k_root
#define PBD_FOLDX_QW 18     // quad-words of a fold x task's extension record (pbd_handle::d_foldx)
#define PBD_FOLD_MAXMIX 8   // fold mode keeps one value per parent mixture / child mixture in registers: K, L <= 8
struct FoldChild {
  const void* sdt[PBD_FOLD_MAXMIX];   // T [H][W]: distance-transformed scores of child mixture k (one pointer per plane: the planes may
                                      // be the child's own response planes, overwritten in place by its y pass)
  uint8_t* ok;                        // output Ik: best child mixture per parent mixture, [L][H][W]
  int K, pad;
  float bias[PBD_FOLD_MAXMIX][PBD_FOLD_MAXMIX];   // bias(k)[m] = biasw[biasid[k] + m] (include/Parts.hpp:172-175), dense: rows beyond K / columns
                                                  // beyond L repeat the last valid one, so the kernel fetches whole rows with wide scalar loads
};
struct FoldJob { int nch, pad; FoldChild ch[PBD_MAX_CH]; };
struct ReduceJob {       // one (level, parent): fold the messages of nch children, in the reference's order
  int H, W, L, nch;
  const void* par_in[PBD_MAX_MIX];   // T: parent mixture m: current score (resp plane or acc slot)
  void* par_out[PBD_MAX_MIX];        // T: parent mixture m: acc slot
  ReduceChild ch[PBD_MAX_CH];        // descending child index (src/DynamicProgram.cpp:95)
};
struct ReduceBlock { int job; unsigned cell0; };  // one 256-thread block of k_reduce
struct RootJob {
  const void* score[PBD_MAX_MIX];  // T: root mixture m current score (entries beyond K repeat mixture K - 1)
  void* rootv; int* rooti;         // rootv: T
  int H, W, K, level, comp;
  float bias;
  unsigned cell0;
  int fold, pad;                   // FoldJob of the root part (score[] are then its raw responses) or -1
};
struct BackLevel {   // per (level, comp) info for backtracking
  const uint8_t* pk;   // best-mixture plane 0 of this comp at this level
  const void* rootv; const int* rooti;   // rootv: T
  int H, W; float scale;
};
struct CandRec { int level, comp, y, x; };

// ---- host model -------------------------------------------------------------
struct PartInfo {
  int comp, p, parent;       // local indices
  int K;                     // #mixtures
  std::vector<int> filterid, defid, biasid;
  std::vector<int> slot;     // acc slot per mixture (global slot id)
  int plane0;                // first pointer plane (global plane id), parent's L planes
  bool leaf;
};

struct pbd_handle {
  // model
  pbd_model_desc md;         // pointers into the vectors below
  std::vector<float> filters, defw, biasw;
  std::vector<int> anchors, part_offset, parentid, mix_offset, filterid, defid, biasid;
  pbd_options opt;
  int max_parts = 0, nslots = 0, nplanes = 0;
  std::vector<char> level_set;   // pbd_set_levels: levels this handle processes (empty = all), intersected with [level_begin, level_end)
  std::vector<PartInfo> parts;                 // flat parts
  std::vector<std::vector<int>> rounds;        // flat part ids whose DT runs in round r
  std::vector<std::vector<std::vector<int>>> red_rounds;  // [round][wave] -> flat child part ids reduced (grouped by parent at plan time)
  std::vector<int> comp_plane0;
  std::string err;
  int conv_mode = PBD_CONV_EXACT;

  // device model
  int ts = 4;                // sizeof(T): 4 = PartsBasedDetector<float>, 8 = PartsBasedDetector<double>
  void* d_wT = nullptr;      // T [kh*kw][flen][nfpad] filters transposed (and converted to T) for the conv kernels
  int ncu = 256;
  int nfpad = 0;
  float* d_biasw = nullptr;
  uint8_t* d_hog_lut = nullptr;   // orientation-snap table of HOGFeatures<T>::features (src/HOGFeatures.cpp:243-249), built once per handle on the device
  int* d_parent = nullptr;   // [ncomp][max_parts] parent of each part
  int* d_plane0 = nullptr;   // [ncomp][max_parts] local plane0 of each part
  int* d_nparts = nullptr;

  // frame plan
  int fw = 0, fh = 0, fcn = 0, nlevels = 0;     // nlevels: levels of ONE frame
  int dt_geom = 0;                               // distance-transform block geometry of float handles: 0 = the measured rule (plan_frame), 1 = 256 lanes / 40 KB, 2 = 128 lanes / 25 KB (pbd_tune_plan)
  int fdepth = 0, fesz = 1;                      // depth of the planned frame's pixels (PBD_DEPTH_*: cv::Mat::depth()), bytes per element
  // A batch of B same-sized frames is planned as B x nlevels "virtual levels" (frame f's level l = f * nlevels + l):
  // every stage is driven by per-level tables, so one launch of a stage then covers all frames of the batch — four
  // times the blocks per launch, the thin rounds of the DP fill the chip and launch tails are paid once per batch.
  int batch = 1, nvl = 0;                       // frames per plan, virtual levels = batch * nlevels
  std::vector<Level> lv;                        // [nvl]
  PyrJob* d_pyrjobs = nullptr;                  // resize jobs, then the pyrDown jobs octave by octave
  struct PyrLaunch { int job0, njobs, maxpix, maxw, maxh; };   // maxpix / maxw / maxh: the largest destination level of the launch
  std::vector<PyrLaunch> pyr_launches;          // [0]: resize, [1..]: pyrDown octave steps
  size_t cells = 0, pyr_bytes = 0;
  bool have_pyr = false, have_feat = false, have_resp = false, have_dp = false;
  // Compact memory plan only: the DP reuses the feature / response memory, so after min() every plane is stale until it
  // is produced (pyramid / pdf) or handed in (pbd_set_level_*) again: have_feat / have_resp are true only when ALL are.
  std::vector<char> feat_ok, resp_ok;            // [level], [level * nfilters + filter]
  // tables handed in by the caller with no min() of this handle behind them: back-tracking needs every plane and every
  // root table of the active levels before it may run
  bool min_ran = false;                          // this plan's tables (Ik, DT pointer planes, roots) come from run_dp_min
  std::vector<char> ext_set, root_set;           // [level * nplanes + plane], [level * ncomp + comp]

  // device frame buffers
  uint8_t* d_img = nullptr; size_t img_cap = 0;
  uint8_t* d_pyr = nullptr;
  char* d_feat = nullptr; char* d_resp = nullptr; char* d_acc = nullptr;   // T data, addressed in bytes (elements * ts)
  uint16_t* d_feat_split = nullptr;   // PBD_CONV_SPLIT: the features as [cell][3 splits][32 channels] bfloat16 (per frame plan)
  bool feat_split_ok = false;         // ... written by k_hog for the features now in d_feat (false: handed in by the caller -> k_feat_split before the bank)
  float* d_split_oscale = nullptr;    // PBD_CONV_SPLIT_F16: [filter] 2^-(12 + e), the responses' scale (e: the filter's weight exponent)
  int split_parts = 0;                // 3: PBD_CONV_SPLIT (bfloat16 parts), 2: PBD_CONV_SPLIT_F16 (binary16 parts), 0: no split bank
  uint16_t* d_wS = nullptr;           // PBD_CONV_SPLIT: the filters as [tap][2 k-steps][3 splits][n-tile][2 k-groups][32][8] bfloat16 (per model)
  uint8_t* d_pk = nullptr;
  unsigned long long* d_scr_base = nullptr;   // [nlevels][nflat parts] element offset of mixture 0's DT planes (ix / iy / sdt)
  std::vector<unsigned long long> scr_base;   // host copy (pbd_get_dp_pointers)
  int* d_flat = nullptr;     // [ncomp][max_parts] flat part index
  int* d_depth = nullptr;    // [ncomp][max_parts] depth of each part in its tree (root = 0)
  int max_depth = 0;
  char* d_rootv = nullptr; int* d_rooti = nullptr;
  char* d_dt_tmpT = nullptr; char* d_dt_sdt = nullptr; int16_t* d_dt_ixT = nullptr; int16_t* d_dt_iy = nullptr;
  size_t dt_cap_elems = 0;
  LevelDev* d_levels = nullptr;
  HogTile* d_hog_tiles = nullptr; int n_hog_tiles = 0; int hog_tc = 16;
  ConvTile* d_conv_tiles = nullptr; int n_conv_tiles = 0;
  // DP tables (all rounds back to back)
  DtMap* d_dtmaps = nullptr; DtTask* d_dttasks = nullptr;   // a task carries its group descriptor
  ReduceJob* d_redjobs = nullptr; ReduceBlock* d_redblocks = nullptr; RootJob* d_rootjobs = nullptr; BackLevel* d_back = nullptr;
  struct ReduceWave { int blk0, nblks; };
  struct RoundLaunch { int xtask0, nxtasks, ytask0, nytasks; size_t lds_x, lds_y; int fold_x; std::vector<ReduceWave> waves;
                       size_t foldx0 = 0; };             // fold x launch: its first record in d_foldx (PBD_FOLDX_QW quad-words per task)
  size_t dt_lds = 0;                                 // LDS budget of a k_dt_pass block in the fullest launch of a frame (thinner launches get less)
  bool unique_filters = false;                       // every filter id belongs to exactly one (component, part, mixture)
  bool compact = false;                              // memory plan of the current frame geometry (plan_frame)
  bool fold = false;                                 // DP structure of this handle: messages folded by the parent's x pass (no k_reduce, no acc planes)
  FoldJob* d_foldjobs = nullptr;
  unsigned long long* d_foldx = nullptr;             // per fold x task, in task order: the part's raw plane pointers [8] + the first child's plane pointers [8] + its Ik base + the number of children: what
                                                     // the block's loader needs for its first loads, at an address that depends on blockIdx only (k_dt_pass fetches it beside the
                                                     // task descriptor instead of behind it)
  int fold_mix = 0;                                  // largest mixture count of a part (the fold kernels' register-array bound)
  int dt_nt = PBD_DT_NT_DEFAULT;                                   // lanes of a k_dt_pass block (64 or 128)
  int dt_nt_x = PBD_DT_NT_DEFAULT;                                 // lanes of a fold x-pass block
  int dt_seg = 0;                                    // target segment length of the DT scans (0: as many lines per block as fit)
  int xcd_chunk = 16;                                // consecutive k_dt_pass tasks kept on one XCD (0: table order)
  std::vector<RoundLaunch> rl;
  int n_rootjobs = 0; unsigned root_cells = 0, root_maxcells = 0;
  int nms_sz = 0;                 // pbd_options.reserved[0]: window of the score-map NMS in front of the back-tracking (0: off, the reference's state)
  uint8_t* d_nms_mask = nullptr;  // [cells * ncomponents]: local maxima of the root planes (same element offsets as d_rootv)
  ReduceBlock* d_rootblocks = nullptr; int n_rootblocks = 0;   // k_root: one 256-thread block per 256 cells of a root job
  // candidates
  int* d_cand_count = nullptr; CandRec* d_cand_rec = nullptr;
  char* d_cand_out = nullptr; char* h_cand_out = nullptr; int* h_cand_count = nullptr;
  bool out_on_host = false;                          // the last back-tracking wrote its records straight into h_cand_out (run_argmin_enqueue)
  size_t cand_stride = 0;
  bool pending = false;
  int first_copy = 0;        // candidate records copied back together with the count (records): starts at PBD_FIRST_COPY per frame of the
                             // plan and grows to 1.25 x the largest count seen, so that the steady state is ONE D2H and no second sync
  char* d_gsend = nullptr;   // set by an RCCL-gathering pbd_group: {count, pad to 16 B, first records} block sent by ncclAllGather

  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool profiling = false;
  hipEvent_t ev[8] = {};
  float stage_ms[6] = {0, 0, 0, 0, 0, 0};
  hipEvent_t ev_dp0 = nullptr, ev_dp1 = nullptr;
  double dp_ms_sum = 0; int dp_frames = 0; bool dp_timer_on = true; bool dp_timed = false;   // DP events are recorded only while profiling
  hipGraphExec_t gexec = nullptr;   // pbd_options.graph: the frame's launches, captured once per geometry
  int frames_on_plan = 0;           // frames enqueued since the last plan_frame
  std::vector<void*> frame_allocs;  // everything freed on re-plan
  size_t frame_bytes = 0, model_bytes = 0;   // device memory held for the frame plan / the model (pbd_get_footprint)
  // pointer tables handed in by the caller (pbd_set_dp_pointers: a DynamicProgram::argmin fed tables that this handle's
  // min() did not produce): composed Ix / Iy per (level, component, plane), allocated on first use; back-tracking
  // reads them instead of the DT planes until the next min()
  int16_t* d_extx = nullptr; int16_t* d_exty = nullptr; unsigned long long* d_ext_base = nullptr;
  bool ext_ptr = false;
  bool root_dirty = false;   // pbd_set_root since the last min(): argmin re-thresholds the root tables first
};

// ---- scalar helpers: the reference's std:: overloads resolve on T ---------------
#ifdef __HIPCC__
__device__ __forceinline__ float t_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double t_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float t_floor(float v) { return floorf(v); }
__device__ __forceinline__ double t_floor(double v) { return floor(v); }
__device__ __forceinline__ float t_fmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double t_fmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ int t_round(float v) { return __float2int_rn(v); }    // cvRound: half to even
__device__ __forceinline__ int t_round(double v) { return __double2int_rn(v); }
#endif

// ---- dynamic-LDS opt-in --------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that drives several
// GPUs (pbd_group, or handles created with different pbd_options.device) must set it on each of them.  One
// instance per kernel instantiation; remembers the largest size every device has been given.
#include <mutex>
struct LdsOptIn {
  std::mutex mu;
  size_t cfg[64] = {};
  hipError_t ensure(const void* fn, size_t lds) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    std::lock_guard<std::mutex> g(mu);
    if (lds <= cfg[d] && cfg[d]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) cfg[d] = lds;
    return e;
  }
};

// ---- probes ----------------------------------------------------------------------
// Per-phase wall-clock stamps and the environment tuning knobs are compiled only into the probe build
// (make probes -> libpbd_hip_probes.so, -DPBD_PROBES), which tests/tools_*.py load; the product library carries
// neither (the pbd_debug_* entry points then return PBD_ERR_UNSUPPORTED).
#if defined(PBD_PROBES) || defined(PBD_TUNE)
#define PBD_PROBE_ENV(name) getenv(name)
#else
#define PBD_PROBE_ENV(name) ((const char*)nullptr)
#endif

// ---- internals shared with pbd_group.cpp -----------------------------------------
int pbd_i_upload_image(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride);   // plan + async H2D
int pbd_i_enqueue_all(pbd_handle* h, const uint8_t* d_src, int stride);                          // all stages + argmin
int pbd_i_collect(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* count);
int pbd_i_finish_frame(pbd_handle* h, int found);
int pbd_i_emit(pbd_handle* h, const std::vector<const char*>& recs, pbd_candidate_head* heads, int32_t* boxes,
               int32_t* locs, int capacity);
#define PBD_FIRST_COPY 192   // candidate records fetched (or gathered) together with the count

// ---- kernel launchers (k_*.hip) ----------------------------------------------
void launch_resize(const PyrJob* jobs, int njobs, int maxpix, int cn, int sstride, const uint8_t* src, uint8_t* pyr, hipStream_t s);
void launch_pyrdown(const PyrJob* jobs, int njobs, int maxw, int maxh, int cn, uint8_t* pyr, hipStream_t s);
void launch_hog(const HogTile* tiles, int ntiles, const LevelDev* levels, const uint8_t* pyr, void* feat, int ts,
                int cn, int sbin, int tc, const uint8_t* binlut, uint16_t* split, int split_parts, int depth, hipStream_t s);   // depth: PBD_DEPTH_* of the level images
size_t hog_lds_bytes(int sbin, int tc, int ts, int bpp = 3);      // bpp: bytes per pixel (channels x element size)
// the image depths beyond 8 bits (k_pyramid.hip): job offsets in bytes, sstride in bytes
void launch_resize_any(const PyrJob* jobs, int njobs, int maxpix, int cn, int depth, int sstride, const uint8_t* src, uint8_t* pyr, hipStream_t s);
void launch_pyrdown_any(const PyrJob* jobs, int njobs, int maxpix, int cn, int depth, uint8_t* pyr, hipStream_t s);
size_t hog_binlut_bytes();                                        // orientation-snap table: best_o for every (dx, dy) in [-255, 255]^2
void launch_hog_binlut(uint8_t* lut, int ts, hipStream_t s);      // evaluated in T (ts = sizeof(T)) with the reference's own chain (k_hog.hip)
// split-product filter bank (k_conv_split.hip): fp32 features -> three exact bfloat16 parts; kh x kw x 32 filters, float responses
void launch_feat_split(const float* feat, uint16_t* out, size_t ncells, hipStream_t s);
void launch_conv_split(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                       float* resp, int nf, int kh, int kw, int variant, hipStream_t s);
void launch_conv_split_persistent(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                                  float* resp, int nf, int ncu, hipStream_t s);   // 5 x 5 banks: persistent workgroups, staging hidden under the MFMAs (tuning variant, not adopted)
void conv_split_filters(const float* filters, int nf, int kh, int kw, std::vector<uint16_t>& out);   // host: the d_wS layout
// PBD_CONV_SPLIT_F16: two scaled binary16 parts per operand, three products (k_conv_split.hip)
void launch_feat_split16(const float* feat, uint16_t* out, size_t ncells, hipStream_t s);
void conv_split16_filters(const float* filters, int nf, int kh, int kw, std::vector<uint16_t>& out, std::vector<float>& oscale);   // oscale[filter]: the response scale
void launch_conv_split16(const ConvTile* tiles, int ntiles, const LevelDev* levels, const uint16_t* feat_split, const uint16_t* wS,
                         float* resp, int nf, int kh, int kw, const float* oscale, int variant, hipStream_t s);
void launch_conv_exact(const ConvTile* tiles, int ntiles, const LevelDev* levels, const void* feat,
                       const void* wT, void* resp, int ts, int nf, int nfpad, int kh, int kw, hipStream_t s);
void launch_conv_mfma(const ConvTile* tiles, int ntiles, const LevelDev* levels, const float* feat,
                      const float* wT, float* resp, int nf, int nfpad, int kh, int kw, hipStream_t s);
void launch_conv_mfma_f64(const ConvTile* tiles, int ntiles, const LevelDev* levels, const double* feat,
                          const double* wT, const double* w4u, double* resp, int nf, int nfpad, int kh, int kw, hipStream_t s);
extern int g_conv_lds_req_kb;
void launch_conv_glds_f32(const ConvTile* tiles, int ntiles, const LevelDev* levels, const float* feat, const float* wT,
                          float* resp, int nf, int nfpad, const float* border, int wg_per_cu, int ncu, hipStream_t s);
void launch_conv_mfma16_f32(const ConvTile* tiles, int ntiles, const LevelDev* levels, const float* feat,
                            const float* wT, const float* w4u, float* resp, int nf, int nfpad, int nhalf, hipStream_t s, int kh, int kw);
void launch_dt_pass(const DtTask* tasks, int ntasks, const DtMap* maps, const FoldJob* folds, const unsigned long long* foldx, const float* biasw, size_t lds,
                    int ts, int nt, int fm, hipStream_t s);
size_t dt_lds_bytes(int stride, int lpb, int ts, int nt);
void launch_reduce(const ReduceJob* jobs, const ReduceBlock* blocks, int nblocks, const float* biasw, int correct_ptr,
                   int ts, hipStream_t s);
void launch_root(const RootJob* jobs, const ReduceBlock* blocks, int nblocks, double thresh, int* count, CandRec* rec,
                 int capacity, int ts, const FoldJob* folds, const float* biasw, int rescan, int fm, const uint8_t* nms_mask,
                 const char* rootv_base, hipStream_t s);
void launch_nms_roots(const RootJob* jobs, int njobs, unsigned maxcells, const char* rootv_base, int ts, int sz, uint8_t* mask, hipStream_t s);
void launch_backtrack(const int* count, const CandRec* rec, int capacity, const BackLevel* back, int ncomp,
                      const int* parent, const int* plane0, const int* nparts, int max_parts, int kh,
                      char* out, size_t out_stride, int ts, const int* flat, const int* depth, int max_depth, int nflat,
                      const unsigned long long* scr_base, const int16_t* ix, const int16_t* iy, int correct_ptr,
                      const int16_t* extx, const int16_t* exty, const unsigned long long* ext_base, int* count_out, hipStream_t s);
void dt_debug_read(unsigned long long* out);
int dt_debug_trace(unsigned long long* t, unsigned* hw, int* nlaunch);   // probe build only
void hog_debug_read(unsigned long long* out);
void conv_debug_read(unsigned long long* out);
void launch_nms_map(const float* src, int rows, int cols, int sz, uint8_t* dst, hipStream_t s);
