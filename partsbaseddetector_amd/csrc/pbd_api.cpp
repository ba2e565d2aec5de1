// pbd_api.cpp — host side of libpbd_hip.so: model ingestion, part-tree round
// scheduling, per-geometry frame plan (buffers + kernel work tables), stage
// orchestration and the C ABI of include/pbd_c.h.
//
// The product path never falls back to a CPU implementation: every stage is a
// HIP kernel launch; host code only plans, launches and (for the few hundred
// candidates of a frame) orders the output like a single-threaded reference run.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include "pbd_internal.hpp"
#include "dt_core.hpp"   // dt_segments: the planner and the kernels share one definition

#define HIPCHK(h, call)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return PBD_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

static int fail(pbd_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

// Kernel launches return nothing: a launch the runtime rejected (wrong current device, a dynamic-LDS request over
// the opt-in, a bad grid) would otherwise leave the previous frame's buffers in place and detect() would return
// stale candidates with PBD_OK.  Checked after every stage.
#define LAUNCHCHK(h, what)                                                               \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) {                                                              \
      (h)->err = std::string(what) + ": kernel launch failed: " + hipGetErrorString(e_); \
      return PBD_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)
// hipGetLastError() is the calling THREAD's sticky last error: an unrelated HIP call of the caller that failed
// earlier (or one of ours whose result was deliberately ignored) would be reported as this frame's launch failure.
// Every entry point that launches clears it first, so LAUNCHCHK only ever sees the library's own launches.
#define CLEAR_STICKY() ((void)hipGetLastError())
// every ABI entry that launches or copies runs on the handle's device, whatever the caller's current device is
#define ON_DEVICE(h) do { HIPCHK(h, hipSetDevice((h)->opt.device)); CLEAR_STICKY(); } while (0)

// ---------------------------------------------------------------------------
// pyramid geometry — HOGFeatures<T>::pyramid, src/HOGFeatures.cpp:98-127,174-175
// ---------------------------------------------------------------------------
static inline int cv_round_f(float v) { return (int)std::lrint((double)v); }

static int compute_geometry(int w, int h, int sbin, int interval, int* nlevels, Level* lv) {
  const float sf = (float)std::pow(2.0, (double)(1.0f / (float)interval));  // HOGFeatures.hpp:78
  const float fw = (float)w, fh = (float)h;
  const float mn = fh < fw ? fh : fw;
  const float r = std::log(mn / (5.0f * (float)sbin)) / std::log(sf);          // :99 (float math)
  const int n = (int)(1.0f + std::floor(r));
  if (n < interval || n > PBD_MAX_LEVELS) return -1;
  for (int i = 0; i < interval; ++i) {
    const float f = (float)(1.0f / std::pow((double)sf, (double)i));            // :116
    lv[i].iw = cv_round_f(fw * f);
    lv[i].ih = cv_round_f(fh * f);
    lv[i].scale = (float)(std::pow((double)sf, (double)i) * (double)sbin);      // :118
    for (int j = i + interval; j < n; j += interval) {
      lv[j].iw = (lv[j - interval].iw + 1) / 2;                                 // :122 pyrDown
      lv[j].ih = (lv[j - interval].ih + 1) / 2;
      lv[j].scale = 2 * lv[j - interval].scale;                                 // :124
    }
  }
  for (int l = 0; l < n; ++l) {
    lv[l].bw = (int)std::round((float)lv[l].iw / (float)sbin);                  // :174
    lv[l].bh = (int)std::round((float)lv[l].ih / (float)sbin);
    lv[l].cw = std::max(lv[l].bw - 2, 0);                                       // :175
    lv[l].ch = std::max(lv[l].bh - 2, 0);
  }
  *nlevels = n;
  return 0;
}

// ---------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------
static int nmix_of(const pbd_handle* h, int fp) { return h->mix_offset[fp + 1] - h->mix_offset[fp]; }

static int ingest_model(pbd_handle* h, const pbd_model_desc* m) {
  if (!m || !m->filters || !m->defw || !m->anchors || !m->biasw || !m->part_offset || !m->parentid ||
      !m->mix_offset || !m->filterid || !m->defid || !m->biasid)
    return fail(h, PBD_ERR_ARG, "model: null pointer");
  if (m->flen != PBD_FLEN || m->norient != PBD_NORIENT)
    return fail(h, PBD_ERR_UNSUPPORTED, "model: only flen=32 / norient=18 HOG is supported");
  if (m->nfilters <= 0 || m->kh <= 0 || m->kw <= 0 || m->kh > 9 || m->kw > 9 || m->sbin <= 0 || m->interval <= 0 ||
      m->interval > 16 || m->ncomponents <= 0)
    return fail(h, PBD_ERR_ARG, "model: bad sizes");
  if (m->ndefs < 0 || m->nbias <= 0) return fail(h, PBD_ERR_ARG, "model: ndefs >= 0 and nbias > 0 required");
  const int nc = m->ncomponents;
  if (m->part_offset[0] != 0) return fail(h, PBD_ERR_ARG, "model: part_offset[0] must be 0");
  for (int c = 0; c < nc; ++c)
    if (m->part_offset[c + 1] <= m->part_offset[c]) return fail(h, PBD_ERR_ARG, "model: part_offset must be strictly increasing");
  h->part_offset.assign(m->part_offset, m->part_offset + nc + 1);
  const int np = h->part_offset[nc];
  if (m->mix_offset[0] != 0) return fail(h, PBD_ERR_ARG, "model: mix_offset[0] must be 0");
  for (int fp = 0; fp < np; ++fp)
    if (m->mix_offset[fp + 1] <= m->mix_offset[fp]) return fail(h, PBD_ERR_ARG, "model: mix_offset must be strictly increasing");
  h->parentid.assign(m->parentid, m->parentid + np);
  h->mix_offset.assign(m->mix_offset, m->mix_offset + np + 1);
  const int nm = h->mix_offset[np];
  h->filterid.assign(m->filterid, m->filterid + nm);
  h->defid.assign(m->defid, m->defid + nm);
  h->biasid.assign(m->biasid, m->biasid + nm);
  h->filters.assign(m->filters, m->filters + (size_t)m->nfilters * m->kh * m->kw * m->flen);
  h->defw.assign(m->defw, m->defw + (size_t)m->ndefs * 4);
  h->anchors.assign(m->anchors, m->anchors + (size_t)m->ndefs * 2);
  h->biasw.assign(m->biasw, m->biasw + m->nbias);
  h->md = *m;
  h->md.filters = h->filters.data(); h->md.defw = h->defw.data(); h->md.anchors = h->anchors.data();
  h->md.biasw = h->biasw.data(); h->md.part_offset = h->part_offset.data(); h->md.parentid = h->parentid.data();
  h->md.mix_offset = h->mix_offset.data(); h->md.filterid = h->filterid.data(); h->md.defid = h->defid.data();
  h->md.biasid = h->biasid.data();

  h->parts.resize(np);
  h->comp_plane0.assign(nc + 1, 0);
  h->max_parts = 0;
  int slot_next = 0, plane_next = 0;
  bool aliasing = false;
  for (int c = 0; c < nc; ++c) {
    const int p0 = h->part_offset[c], cnp = h->part_offset[c + 1] - p0;
    if (cnp <= 0) return fail(h, PBD_ERR_ARG, "model: empty component");
    if (cnp > 256) return fail(h, PBD_ERR_UNSUPPORTED, "model: more than 256 parts in a component");   // BT_MAXP (k_backtrack)
    h->max_parts = std::max(h->max_parts, cnp);
    h->comp_plane0[c] = plane_next;
    std::map<int, int> slot_of;   // filter id -> slot (ncscores is indexed by filter id, DynamicProgram.cpp:93)
    std::map<int, int> uses;
    std::vector<int> nchild(cnp, 0);
    for (int p = 1; p < cnp; ++p) {
      const int par = h->parentid[p0 + p];
      if (par < 0 || par >= p) return fail(h, PBD_ERR_ARG, "model: parts must be ordered parent < child");
      nchild[par]++;
    }
    for (int p = 0; p < cnp; ++p) {
      PartInfo& P = h->parts[p0 + p];
      P.comp = c; P.p = p; P.parent = (p == 0) ? -1 : h->parentid[p0 + p];
      P.K = nmix_of(h, p0 + p);
      if (P.K <= 0 || P.K > PBD_MAX_MIX) return fail(h, PBD_ERR_UNSUPPORTED, "model: 1..16 mixtures per part");
      P.leaf = (nchild[p] == 0);
      const int fm0 = h->mix_offset[p0 + p];
      for (int m2 = 0; m2 < P.K; ++m2) {
        const int fid = h->filterid[fm0 + m2];
        if (fid < 0 || fid >= m->nfilters) return fail(h, PBD_ERR_ARG, "model: filterid out of range");
        P.filterid.push_back(fid);
        P.defid.push_back(h->defid[fm0 + m2]);
        P.biasid.push_back(h->biasid[fm0 + m2]);
        if (!slot_of.count(fid)) slot_of[fid] = slot_next++;
        P.slot.push_back(slot_of[fid]);
        if (++uses[fid] > 1) aliasing = true;
        if (p > 0) {
          const int did = h->defid[fm0 + m2];
          if (did < 0 || did >= m->ndefs) return fail(h, PBD_ERR_ARG, "model: defid out of range");
          if (h->defw[did * 4] == 0.f || h->defw[did * 4 + 2] == 0.f)
            return fail(h, PBD_ERR_ARG, "model: quadratic deformation weights must be non-zero "
                                        "(include/DistanceTransform.hpp:99 divides by 2a)");
        }
        const int bid = h->biasid[fm0 + m2];
        const int L = (p == 0) ? 1 : nmix_of(h, p0 + h->parentid[p0 + p]);
        if (bid < 0 || bid + L > m->nbias) return fail(h, PBD_ERR_ARG, "model: biasid out of range");
      }
      P.plane0 = -1;
      if (p > 0) {
        P.plane0 = plane_next;
        plane_next += nmix_of(h, p0 + P.parent);
      }
    }
  }
  h->comp_plane0[nc] = plane_next;
  h->nslots = slot_next;
  h->nplanes = plane_next;

  // ---- round schedule -------------------------------------------------------
  // DT of a part runs once all its children have sent their message (DynamicProgram.cpp:95 walks
  // p = P-1..1 with parent < child): round = height of the part.  Messages into one parent are
  // float adds in DESCENDING child order (:156); to keep those bits, a child's message is folded
  // no earlier than every higher-indexed sibling's (reduce round = max over them), and siblings
  // folded in the same round go through ONE reduce job that adds them in that order.
  // fold mode (messages folded by the consumer, no accumulated planes): needs every part's accumulator to be its own
  // (no filter id shared inside a component: the reference's ncscores is indexed by FILTER id, so two parts with
  // one id would share an accumulator) and its mixtures / its children's to fit the register arrays of the fold
  {
    std::vector<int> fuse(m->nfilters, 0);
    h->unique_filters = true;
    for (int fm = 0; fm < nm; ++fm) if (++fuse[h->filterid[fm]] > 1) h->unique_filters = false;   // (also across components: face-like models share a pool)
  }
  h->fold = !aliasing && h->opt.reserved[1] != 1;
  if (const char* e = PBD_PROBE_ENV("PBD_DP_MODE")) h->fold = h->fold && atoi(e) != 1;   // A/B: the three-kernel structure
  {
    std::vector<int> nchild_flat(np, 0);
    for (int fp = 0; fp < np; ++fp) {
      if (h->parts[fp].K > PBD_FOLD_MAXMIX) h->fold = false;
      h->fold_mix = std::max(h->fold_mix, h->parts[fp].K);
      if (h->parts[fp].p > 0 && ++nchild_flat[h->part_offset[h->parts[fp].comp] + h->parts[fp].parent] > PBD_MAX_CH) h->fold = false;
    }
  }
  h->rounds.clear();
  h->red_rounds.clear();
  if (aliasing) {  // shared filter ids inside a component: keep the reference's strictly sequential order
    for (int c = 0; c < nc; ++c)
      for (int p = h->part_offset[c + 1] - h->part_offset[c] - 1; p > 0; --p) {
        h->rounds.push_back(std::vector<int>(1, h->part_offset[c] + p));
        h->red_rounds.push_back({std::vector<int>(1, h->part_offset[c] + p)});
      }
  } else {
    std::vector<int> height(np, 0), rround(np, 0);
    int nrounds = 0;
    for (int c = 0; c < nc; ++c) {
      const int p0 = h->part_offset[c], cnp = h->part_offset[c + 1] - p0;
      for (int p = cnp - 1; p > 0; --p) {  // children before parents (parent < child)
        const int par = h->parts[p0 + p].parent;
        height[p0 + par] = std::max(height[p0 + par], height[p0 + p] + 1);
      }
      for (int p = cnp - 1; p > 0; --p) {  // descending index: higher siblings first
        int rr = height[p0 + p];
        for (int q = p + 1; q < cnp; ++q)
          if (h->parts[p0 + q].parent == h->parts[p0 + p].parent) rr = std::max(rr, rround[p0 + q]);
        rround[p0 + p] = rr;
        nrounds = std::max(nrounds, rr + 1);
      }
    }
    h->rounds.assign(nrounds, {});
    std::vector<std::vector<int>> red(nrounds);
    for (int fp = 0; fp < np; ++fp) {
      if (h->parts[fp].p == 0) continue;
      h->rounds[height[fp]].push_back(fp);
      red[rround[fp]].push_back(fp);
    }
    // waves: at most PBD_MAX_CH children of one parent per reduce job; overflow goes to a later wave
    h->red_rounds.assign(nrounds, {});
    for (int r = 0; r < nrounds; ++r) {
      std::vector<int> rest(red[r].rbegin(), red[r].rend());  // descending flat index
      while (!rest.empty()) {
        std::vector<int> wave, next;
        std::map<int, int> cnt;
        for (int fp : rest) {
          const int parent_fp = h->part_offset[h->parts[fp].comp] + h->parts[fp].parent;
          if (cnt[parent_fp] < PBD_MAX_CH && !std::count_if(next.begin(), next.end(), [&](int g) {
                return h->part_offset[h->parts[g].comp] + h->parts[g].parent == parent_fp; })) {
            cnt[parent_fp]++;
            wave.push_back(fp);
          } else {
            next.push_back(fp);
          }
        }
        h->red_rounds[r].push_back(wave);
        rest.swap(next);
      }
    }
  }
  return PBD_OK;
}

static int upload_model(pbd_handle* h) {
  const pbd_model_desc& m = h->md;
  // filters transposed to [tap][c][nfpad] (n contiguous): scalar loads in the VALU kernel,
  // B-operand rows in the MFMA kernel.  nfpad is a multiple of 160 (5 x 32-wide MFMA n-tiles).
  h->nfpad = ((m.nfilters + 159) / 160) * 160;
  // + one trailing border cell (0, and 1 in the truncation channel flen - 1): the source the persistent filter-bank kernel
  // streams out-of-level cells from (src/SpatialConvolutionEngine.cpp:147-155)
  const size_t wt_n = (size_t)m.kh * m.kw * m.flen * h->nfpad;
  std::vector<float> wT(3 * wt_n + m.flen, 0.f);
  wT[wt_n + m.flen - 1] = 1.f;
  // ... and the same filters once more as [tap][16-channel half][k = 0..3][nfpad][s = 0..3] = channel 16 half + 4 k + s: the lane
  // (k, filter) of the persistent kernel's B operand reads its four k-steps of a tap with ONE 16-byte load (eight
  // global_load_dword per 32 MFMAs cost the MFMA pipe a quarter of its rate: tests/tools/mfma_rate_probe.hip)
  if (m.flen == PBD_FLEN)
    for (int n = 0; n < m.nfilters; ++n)
      for (int tap = 0; tap < m.kh * m.kw; ++tap)
        for (int c = 0; c < m.flen; ++c)
        {
          const float w = h->filters[((size_t)n * m.kh * m.kw + tap) * m.flen + c];
          wT[wt_n + m.flen + ((((size_t)tap * 2 + c / 16) * 4 + (c % 16) / 4) * h->nfpad + n) * 4 + c % 4] = w;     // k = (c % 16) / 4, s = c % 4
          // third copy, [tap][half][k][nfpad][u] with channel = 16 half + 4 u + k: the B operand of k_conv_mfma16<.., B4>
          wT[2 * wt_n + m.flen + ((((size_t)tap * 2 + c / 16) * 4 + c % 4) * h->nfpad + n) * 4 + (c % 16) / 4] = w;
        }
  for (int n = 0; n < m.nfilters; ++n)
    for (int i = 0; i < m.kh; ++i)
      for (int j = 0; j < m.kw; ++j)
        for (int c = 0; c < m.flen; ++c)
          wT[((size_t)(i * m.kw + j) * m.flen + c) * h->nfpad + n] =
              h->filters[(((size_t)n * m.kh + i) * m.kw + j) * m.flen + c];
  HIPCHK(h, hipMalloc(&h->d_wT, wT.size() * h->ts));
  if (h->ts == 8) {   // filters convertTo(DataType<double>), src/PartsBasedDetector.cpp:113-117 (exact widening)
    std::vector<double> wd(wT.begin(), wT.end());
    // the double filter bank's 16-byte B layout replaces the (unused) float copy behind the border cell:
    // [tap][8-channel group][k][nfpad][u = 0, 1], channel = 8 group + 4 u + k (k_conv_mfma16<double, 4, .., B4>)
    if (m.flen == PBD_FLEN)
      for (int n = 0; n < m.nfilters; ++n)
        for (int tap = 0; tap < m.kh * m.kw; ++tap)
          for (int c = 0; c < m.flen; ++c)
            wd[wt_n + m.flen + ((((size_t)tap * 4 + c / 8) * 4 + c % 4) * h->nfpad + n) * 2 + (c % 8) / 4] =
                (double)h->filters[((size_t)n * m.kh * m.kw + tap) * m.flen + c];
    HIPCHK(h, hipMemcpy(h->d_wT, wd.data(), wd.size() * sizeof(double), hipMemcpyHostToDevice));
  } else {
    HIPCHK(h, hipMemcpy(h->d_wT, wT.data(), wT.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  size_t split_bytes = 0;
  h->split_parts = h->conv_mode == PBD_CONV_SPLIT ? 3 : h->conv_mode == PBD_CONV_SPLIT_F16 ? 2 : 0;
  if (h->split_parts) {   // the three exact bfloat16 parts of every weight (or two binary16 parts of the scaled weight), in the MFMA operand order of k_conv_split32
    std::vector<uint16_t> wS;
    if (h->split_parts == 3) conv_split_filters(h->filters.data(), m.nfilters, m.kh, m.kw, wS);
    else {
      std::vector<float> osc;
      conv_split16_filters(h->filters.data(), m.nfilters, m.kh, m.kw, wS, osc);
      HIPCHK(h, hipMalloc((void**)&h->d_split_oscale, osc.size() * sizeof(float)));
      HIPCHK(h, hipMemcpy(h->d_split_oscale, osc.data(), osc.size() * sizeof(float), hipMemcpyHostToDevice));
      split_bytes += osc.size() * sizeof(float);
    }
    split_bytes += wS.size() * sizeof(uint16_t);
    HIPCHK(h, hipMalloc((void**)&h->d_wS, wS.size() * sizeof(uint16_t)));
    HIPCHK(h, hipMemcpy(h->d_wS, wS.data(), wS.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  // orientation-snap table of the HOG kernel (k_hog.hip): 511 x 511 bytes, computed on the device by the reference's own
  // comparison chain in T
  HIPCHK(h, hipMalloc(&h->d_hog_lut, hog_binlut_bytes()));
  launch_hog_binlut(h->d_hog_lut, h->ts, h->stream);
  LAUNCHCHK(h, "HOG orientation table");
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // biasw plus one trailing 0 (used by the stand-alone pbd_dt2d)
  std::vector<float> bw(h->biasw);
  bw.push_back(0.f);
  HIPCHK(h, hipMalloc(&h->d_biasw, bw.size() * sizeof(float)));
  HIPCHK(h, hipMemcpy(h->d_biasw, bw.data(), bw.size() * sizeof(float), hipMemcpyHostToDevice));
  const int nc = m.ncomponents, mp = h->max_parts;
  std::vector<int> par(nc * mp, 0), pl0(nc * mp, 0), npv(nc, 0), flt(nc * mp, 0), dep(nc * mp, 0);
  h->max_depth = 0;
  for (int c = 0; c < nc; ++c) {
    const int p0 = h->part_offset[c], cnp = h->part_offset[c + 1] - p0;
    npv[c] = cnp;
    for (int p = 0; p < cnp; ++p) {
      par[c * mp + p] = h->parts[p0 + p].parent;
      pl0[c * mp + p] = (p > 0) ? h->parts[p0 + p].plane0 - h->comp_plane0[c] : 0;
      flt[c * mp + p] = p0 + p;
      dep[c * mp + p] = (p > 0) ? dep[c * mp + h->parts[p0 + p].parent] + 1 : 0;   // parents precede children (validated)
      h->max_depth = std::max(h->max_depth, dep[c * mp + p]);
    }
  }
  HIPCHK(h, hipMalloc(&h->d_flat, flt.size() * sizeof(int)));
  HIPCHK(h, hipMalloc(&h->d_depth, dep.size() * sizeof(int)));
  HIPCHK(h, hipMemcpy(h->d_flat, flt.data(), flt.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_depth, dep.data(), dep.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(h, hipMalloc(&h->d_parent, par.size() * sizeof(int)));
  HIPCHK(h, hipMalloc(&h->d_plane0, pl0.size() * sizeof(int)));
  HIPCHK(h, hipMalloc(&h->d_nparts, npv.size() * sizeof(int)));
  HIPCHK(h, hipMemcpy(h->d_parent, par.data(), par.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_plane0, pl0.data(), pl0.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_nparts, npv.data(), npv.size() * sizeof(int), hipMemcpyHostToDevice));
  // candidates
  const int cap = h->opt.max_candidates;
  h->cand_stride = sizeof(pbd_candidate_head) + (size_t)mp * 28;
  HIPCHK(h, hipMalloc(&h->d_cand_count, sizeof(int)));
  HIPCHK(h, hipMalloc(&h->d_cand_rec, sizeof(CandRec) * cap));
  HIPCHK(h, hipMalloc(&h->d_cand_out, h->cand_stride * cap));
  HIPCHK(h, hipHostMalloc((void**)&h->h_cand_out, h->cand_stride * cap));
  HIPCHK(h, hipHostMalloc((void**)&h->h_cand_count, sizeof(int) * 4));
  h->model_bytes = wT.size() * h->ts + hog_binlut_bytes() + bw.size() * sizeof(float) + (par.size() * 4 + npv.size()) * sizeof(int) + sizeof(int) +
                   sizeof(CandRec) * cap + h->cand_stride * cap + split_bytes;
  return PBD_OK;
}

// ---------------------------------------------------------------------------
// frame plan
// ---------------------------------------------------------------------------
static const int kFirstCopy = PBD_FIRST_COPY;  // records fetched together with the count (per frame of the plan; grows, pbd_i_finish_frame)
template <typename T>
static int dev_alloc(pbd_handle* h, T** p, size_t n) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return PBD_ERR_HIP; }
  h->frame_allocs.push_back(q);
  h->frame_bytes += std::max<size_t>(n, 1) * sizeof(T);
  *p = (T*)q;
  return PBD_OK;
}
template <typename T>
static int dev_upload(pbd_handle* h, T** p, const std::vector<T>& v) {
  int rc = dev_alloc(h, p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(h, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return PBD_OK;
}

static void free_frame(pbd_handle* h) {
  if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }   // the captured launches point into the buffers freed below
  h->frames_on_plan = 0;
  for (void* p : h->frame_allocs) hipFree(p);
  h->frame_allocs.clear();
  h->frame_bytes = 0;
  h->d_extx = h->d_exty = nullptr; h->d_ext_base = nullptr; h->ext_ptr = false;
  h->fw = h->fh = h->fcn = 0; h->fdepth = 0; h->fesz = 1;
  h->have_pyr = h->have_feat = h->have_resp = h->have_dp = false;
  h->min_ran = false;
  h->feat_split_ok = false;
  h->feat_ok.clear(); h->resp_ok.clear(); h->ext_set.clear(); h->root_set.clear();
}

// Compact plan: validity of the stage planes the DP overwrites (pbd_internal.hpp).  all = true: a producer stage has just
// written every plane; false: min() has just reused the memory.
static void compact_mark_feat(pbd_handle* h, bool all) { if (h->compact) h->feat_ok.assign((size_t)h->nvl, all ? 1 : 0); }
static void compact_mark_resp(pbd_handle* h, bool all) { if (h->compact) h->resp_ok.assign((size_t)h->nvl * h->md.nfilters, all ? 1 : 0); }
static bool all_active_set(const pbd_handle* h, const std::vector<char>& v, int per_level) {
  if (v.size() != (size_t)h->nvl * per_level) return false;
  for (int l = 0; l < h->nvl; ++l) {
    if (!h->lv[l].active || h->lv[l].cw == 0 || h->lv[l].ch == 0) continue;
    for (int k = 0; k < per_level; ++k) if (!v[(size_t)l * per_level + k]) return false;
  }
  return true;
}

// DT block geometry under an LDS budget.  stride = LDS elements per line: >= len + 1 and ODD — the (y, z) pairs of
// element e of consecutive lines are then 2 * (stride mod 32) banks apart instead of in the same banks (lanes of
// different lines work on similar element indices at the same time: with an even stride of 160 every LDS access
// of the scan was an lpb-way bank conflict); lpb = lines per block: 4 .. lanes of the block (plain), or a whole
// number of rows x the K mixtures of the part (fold: unit = K).
static int dt_stride_for(int len) { return (len + 1) | 1; }
static int dt_lpb_for(int stride, int len, int unit, size_t budget, int ts, int nt, int seg, bool round_lanes) {
  const int lmin = unit > 1 ? unit : 4;
  int lpb = std::min(nt, 128);   // at most one line per lane
  if (unit > 1) lpb = std::max(unit, lpb / unit * unit);
  while (lpb > lmin && dt_lds_bytes(stride, lpb, ts, nt) > budget) lpb -= (unit > 1 ? unit : 1);
  // plain: the nt / lpb lanes of a line are a whole number, so 45 lines that fit would leave 128 - 2 * 45 lanes idle and
  // every line with two segments where 42 lines get three: the largest lpb <= the fit that uses all lanes
  if (round_lanes && unit <= 1 && lpb > lmin) lpb = std::max(lmin, nt / ((nt + lpb - 1) / lpb));
  // The nt / lpb lanes that share a line scan one segment of it each (dt_core.hpp), and a block lasts as long as
  // its segments are: with a target segment length, lines are given up for lanes per line where the budget
  // would put so many lines into a block that each is left with one or two lanes.
  if (seg > 0) {
    const int P = std::max(1, std::min(nt / 4, (len + seg - 1) / seg));
    int cap = std::max(lmin, nt / P);
    if (unit > 1) cap = std::max(unit, cap / unit * unit);
    lpb = std::min(lpb, cap);
  }
  return lpb;
}
// fold >= 0: the group is one part at one level, a block = whole rows of its nmaps mixtures
// round_lanes: plain groups only — the largest lines-per-block <= the fit that leaves no lane idle (a per-launch choice of plan_frame)
static DtGroup dt_group(int map0, int nmaps, int nlines, int len, size_t budget, int ts, int nt, int seg, int fold = -1, bool round_lanes = true) {
  DtGroup g{};
  g.map0 = map0; g.nmaps = nmaps; g.nlines = nlines; g.len = len; g.fold = fold;
  g.stride = dt_stride_for(len);
  g.lpb = dt_lpb_for(g.stride, len, fold >= 0 ? nmaps : 1, budget, ts, nt, seg, round_lanes);
  // wave-uniform quotients of the block's index arithmetic, as constants (pbd_internal.hpp)
  g.nsub = nt / g.lpb;
  g.P = dt_segments(g.nsub, len);
  g.chunk = (len + g.nsub - 1) / g.nsub;
  g.magic_lpb = dt_magic((unsigned)g.lpb);
  g.magic_nlines = dt_magic((unsigned)nlines);
  g.magic_P = dt_magic((unsigned)g.P);
  return g;
}
// maps: the descriptor table the group indexes (plain groups: a block whose lines are contiguous in memory gets their address, DtTask::src0)
static void dt_add_tasks(const DtGroup& g, std::vector<DtTask>& out, const std::vector<DtMap>* maps = nullptr, int ts = 4) {
  if (g.fold >= 0) {
    const int R = g.lpb / g.nmaps;
    for (int r0 = 0; r0 < g.nlines; r0 += R) out.push_back(DtTask{r0, std::min(R, g.nlines - r0) * g.nmaps, 0, r0, g, nullptr});
  } else {
    const int total = g.nmaps * g.nlines;
    const size_t map_bytes = (size_t)g.nlines * g.len * ts;
    for (int g0 = 0; g0 < total; g0 += g.lpb) {
      DtTask t{g0, std::min(g.lpb, total - g0), g0 / g.nlines, g0 % g.nlines, g, nullptr};
      if (maps && g.len > 1) {
        const int mlast = (g0 + t.nl - 1) / g.nlines;
        bool contig = true;
        for (int m = t.m0; m < mlast && contig; ++m)
          contig = (const char*)(*maps)[(size_t)g.map0 + m + 1].src == (const char*)(*maps)[(size_t)g.map0 + m].src + map_bytes;
        if (contig) t.src0 = (const char*)(*maps)[(size_t)g.map0 + t.m0].src + (size_t)t.l0 * g.len * ts;
      }
      out.push_back(t);
    }
  }
}
// DtGroup::fused (dt_core.hpp: dt_isect's FUSED form, the read-out's fused sum): float maps whose a and b are converted floats — the model's
// weights always (dt_map), pbd_dt2d's caller may hand in any double — on lines short enough for the products to be exact in fp64
static void dt_mark_fused(std::vector<DtTask>& tasks, const DtMap* maps, int ts) {   // (and the group's pointer layout: DT_G_NATURAL)
  for (DtTask& t : tasks) {
    DtGroup& g = t.g;
    bool ok = ts == 4 && g.len <= DT_FUSE_MAXLEN;
    for (int m = 0; m < g.nmaps && ok; ++m) {
      const DtMap& mp = maps[g.map0 + m];
      ok = (double)(float)mp.a == mp.a && (double)(float)mp.b == mp.b && (long long)g.len + std::abs((long long)mp.os) <= DT_FUSE_MAXLEN;
    }
    // (the pointer layout is a property of the pass: every map of a group has the same)
    g.fused = (ok ? DT_G_FUSED : 0) | (g.nmaps > 0 && maps[g.map0].ptr_natural ? DT_G_NATURAL : 0);
  }
}
static DtMap dt_map(const void* src, void* dst, int16_t* ptr, float wq, float wl, int os, int natural) {
  DtMap m{};
  m.src = src; m.dst = dst; m.ptr = ptr;
  m.a = -(double)wq; m.b = -(double)wl;      // Quadratic fx(-w0, -w1), fy(-w2, -w3) (src/DynamicProgram.cpp:125-127)
  m.r2a = 1.0 / (2.0 * m.a);                 // IEEE division (dt_core.hpp: dt_isect)
  m.os = os; m.ptr_natural = natural;
  return m;
}

static int depth_esz(int depth) { return depth == PBD_DEPTH_8U ? 1 : depth == PBD_DEPTH_16U ? 2 : depth == PBD_DEPTH_32F ? 4 : depth == PBD_DEPTH_64F ? 8 : 0; }
static int plan_frame(pbd_handle* h, int w, int hgt, int cn, int batch = 1, int depth = PBD_DEPTH_8U) {
  if (h->fw == w && h->fh == hgt && h->fcn == cn && h->batch == batch && h->fdepth == depth) return PBD_OK;
  if (cn != 1 && cn != 3) return fail(h, PBD_ERR_UNSUPPORTED, "image: 1 or 3 channels");
  const int esz = depth_esz(depth);
  if (!esz) return fail(h, PBD_ERR_UNSUPPORTED, "Unsupported image type (src/HOGFeatures.cpp:136-146: CV_8U, CV_16U, CV_32F, CV_64F)");
  if (esz > 1 && batch != 1) return fail(h, PBD_ERR_UNSUPPORTED, "batches of frames: 8-bit images");
  if (batch < 1 || batch > 64) return fail(h, PBD_ERR_ARG, "batch: 1..64 frames");
  hipStreamSynchronize(h->stream);
  free_frame(h);
  const pbd_model_desc& m = h->md;
  int n = 0;
  h->lv.assign((size_t)PBD_MAX_LEVELS * batch, Level{});
  if (w < 3 || hgt < 3 || compute_geometry(w, hgt, m.sbin, m.interval, &n, h->lv.data()))
    return fail(h, PBD_ERR_ARG, "image too small: the pyramid needs at least `interval` levels "
                                "(src/HOGFeatures.cpp:99,114)");
  const int n1 = n;             // levels of one frame
  h->nlevels = n1; h->batch = batch; h->nvl = n1 * batch;
  h->first_copy = kFirstCopy * batch;
  h->lv.resize(h->nvl);
  for (int f = 1; f < batch; ++f)
    for (int l = 0; l < n1; ++l) h->lv[f * n1 + l] = h->lv[l];
  n = h->nvl;                   // from here on `n` counts the virtual levels (frame f's level l = f * n1 + l)
  int lb = h->opt.level_begin, le = h->opt.level_end;
  if (le <= 0 || le > n1) le = n1;
  if (lb < 0) lb = 0;
  size_t cells = 0, pyr = 0;
  for (int vl = 0; vl < n; ++vl) {
    Level& L = h->lv[vl];
    const int l = vl % n1;
    L.active = (l >= lb && l < le) && (h->level_set.empty() || (l < (int)h->level_set.size() && h->level_set[l]));
    if (L.cw > 32767 || L.ch > 32767) return fail(h, PBD_ERR_UNSUPPORTED, "level too large for 16-bit pointers");
    // the fold loader addresses a level's planes with 32-bit offsets: cell * sizeof(T) and plane * cells + cell (<= 8 planes of a child)
    if ((size_t)L.cw * L.ch >= ((size_t)1 << 28)) return fail(h, PBD_ERR_UNSUPPORTED, "level too large (2^28 cells)");
    L.img_off = pyr; pyr += (size_t)L.iw * L.ih * cn * esz;
    L.cell_off = cells; cells += (size_t)L.cw * L.ch;
  }
  h->cells = cells; h->pyr_bytes = pyr;
  if (cells >= (1u << 31)) return fail(h, PBD_ERR_UNSUPPORTED, "frame too large");
  int rc;
  if ((rc = dev_alloc(h, &h->d_img, (size_t)w * hgt * cn * esz * batch))) return rc;
  {  // image pyramid jobs: the first octave of every frame from the frame (tightly packed, back to back), then the chains
    std::vector<PyrJob> jobs;
    h->pyr_launches.clear();
    pbd_handle::PyrLaunch R{0, 0, 1, 1, 1};
    for (int f = 0; f < batch; ++f)
      for (int i = 0; i < m.interval; ++i) {
        const Level& L = h->lv[f * n1 + i];
        jobs.push_back(PyrJob{(unsigned long long)f * w * hgt * cn * esz, (unsigned long long)L.img_off, w, hgt, L.iw, L.ih});
        R.maxpix = std::max(R.maxpix, L.iw * L.ih);
      }
    R.njobs = (int)jobs.size();
    h->pyr_launches.push_back(R);
    for (int base = m.interval; base < n1; base += m.interval) {
      pbd_handle::PyrLaunch D{(int)jobs.size(), 0, 1, 1, 1};
      for (int f = 0; f < batch; ++f)
        for (int j = base; j < std::min(base + m.interval, n1); ++j) {
          const Level &S = h->lv[f * n1 + j - m.interval], &L = h->lv[f * n1 + j];
          jobs.push_back(PyrJob{(unsigned long long)S.img_off, (unsigned long long)L.img_off, S.iw, S.ih, L.iw, L.ih});
          D.maxpix = std::max(D.maxpix, L.iw * L.ih);
          D.maxw = std::max(D.maxw, L.iw); D.maxh = std::max(D.maxh, L.ih);
        }
      D.njobs = (int)jobs.size() - D.job0;
      h->pyr_launches.push_back(D);
    }
    if ((rc = dev_upload(h, &h->d_pyrjobs, jobs))) return rc;
  }
  const size_t ts = (size_t)h->ts;   // sizeof(T); T buffers are char* addressed as elements * ts
  if ((rc = dev_alloc(h, &h->d_resp, cells * m.nfilters * ts))) return rc;
  // Memory plan.  Default: every stage buffer has its own allocation and stays valid after detect() (the parity
  // tests read features / responses / tables of a finished frame).  Compact (fold structure, every filter id used by
  // one mixture only; chosen automatically for large frames — the responses alone over 400 MB, e.g. 1920x1080 —
  // or forced with dp_mode 2): buffers that are never live together share memory:
  //   * a mixture's distance-transformed scores overwrite its own raw response plane (its x pass has consumed the
  //     plane before its y pass writes it; nothing else reads the raw plane of a non-root part);
  //   * the level images + HOG features (dead once the filter bank has run) share one region with the x pass's
  //     per-round output + the Ik planes (first written by the DP);
  // 1920x1080, person model: 1.47 GB instead of 3.3 GB.  After min() the image / feature / response getters of a
  // compact handle answer PBD_ERR_STATE (the buffers have been reused).
  size_t act_cells0 = 0;
  for (int l = 0; l < n; ++l) if (h->lv[l].active) act_cells0 += (size_t)h->lv[l].cw * h->lv[l].ch;
  size_t maxK0 = 1;
  for (auto& rnd : h->rounds) { size_t k = 0; for (int fp : rnd) k += h->parts[fp].K; maxK0 = std::max(maxK0, k); }
  h->compact = h->fold && h->unique_filters && (h->opt.reserved[1] == 2 || (h->opt.reserved[1] == 0 && cells * m.nfilters * ts > ((size_t)400 << 20)));
  const size_t pk_bytes = cells * std::max(h->nplanes, 1), feat_bytes = cells * PBD_FLEN * ts;
  const size_t al = 256, pyr_al = (pyr + al - 1) / al * al, pk_al = (pk_bytes + al - 1) / al * al;
  if (h->compact) {
    char* u = nullptr;
    if ((rc = dev_alloc(h, &u, std::max(pyr_al + feat_bytes, pk_al + maxK0 * act_cells0 * ts)))) return rc;
    h->d_pyr = (uint8_t*)u; h->d_feat = u + pyr_al;
    h->d_pk = (uint8_t*)u; h->d_dt_tmpT = u + pk_al;
  } else {
    if ((rc = dev_alloc(h, &h->d_pyr, pyr))) return rc;
    if ((rc = dev_alloc(h, &h->d_feat, feat_bytes))) return rc;
    if ((rc = dev_alloc(h, &h->d_pk, pk_bytes))) return rc;
  }
  h->d_feat_split = nullptr;   // (PBD_CONV_SPLIT: placed below, once the DT planes are known — the compact plan shares their memory)
  if ((rc = dev_alloc(h, &h->d_rootv, cells * m.ncomponents * ts))) return rc;
  if ((rc = dev_alloc(h, &h->d_rooti, cells * m.ncomponents))) return rc;
  if (h->nms_sz > 0 && (rc = dev_alloc(h, &h->d_nms_mask, cells * m.ncomponents))) return rc;

  std::vector<LevelDev> ld(n);
  for (int l = 0; l < n; ++l) {
    const Level& L = h->lv[l];
    ld[l] = LevelDev{L.iw, L.ih, L.bw, L.bh, L.cw, L.ch, (unsigned long long)L.img_off, (unsigned long long)L.cell_off};
  }
  if ((rc = dev_upload(h, &h->d_levels, ld))) return rc;

  // HOG tiles: TC x TC cells; shrink the tile until its LDS footprint fits
  h->hog_tc = 16;
  const int hog_bpp = esz == 1 ? 3 : cn * esz;    // (8-bit frames: the tile side does not depend on the channel count)
  if (const char* e = PBD_PROBE_ENV("PBD_HOG_TC")) h->hog_tc = std::max(2, std::min(16, atoi(e)));   // tuning builds
  while (h->hog_tc > 2 && hog_lds_bytes(m.sbin, h->hog_tc, h->ts, hog_bpp) > 150 * 1024) h->hog_tc /= 2;
  if (hog_lds_bytes(m.sbin, h->hog_tc, h->ts, hog_bpp) > 150 * 1024) return fail(h, PBD_ERR_UNSUPPORTED, "sbin too large");
  std::vector<HogTile> ht;
  std::vector<ConvTile> ct;
  for (int l = 0; l < n; ++l) {
    const Level& L = h->lv[l];
    if (!L.active || L.cw == 0 || L.ch == 0) continue;
    for (int y = 0; y < L.ch; y += h->hog_tc)
      for (int x = 0; x < L.cw; x += h->hog_tc) ht.push_back(HogTile{l, y, x, 0});
    for (int y = 0; y < L.ch; y += 16)
      for (int x = 0; x < L.cw; x += 16) ct.push_back(ConvTile{l, y, x, 0});
  }
  // The filter bank runs tile position 8 g + x of this list on XCD x (k_conv.hip: groups of 8 tiles, all n-tiles of a
  // tile on one XCD).  Horizontally adjacent tiles write the two halves of the same 128-byte lines of every response
  // plane (a tile row is 64 bytes); in list order they sat on DIFFERENT XCDs, whose L2s cannot merge the halves
  // (WRITE_SIZE 124 MB for 88 MB of responses).  Pairs of neighbours (2k, 2k + 1) go to the same XCD, one group apart.
  {
    std::vector<ConvTile> o(ct);
    const size_t full = ct.size() / 16 * 16;
    for (size_t b = 0; b < full; b += 16)
      for (size_t x = 0; x < 8; ++x) { o[b + x] = ct[b + 2 * x]; o[b + 8 + x] = ct[b + 2 * x + 1]; }
    ct.swap(o);
  }
  h->n_hog_tiles = (int)ht.size();
  h->n_conv_tiles = (int)ct.size();
  if ((rc = dev_upload(h, &h->d_hog_tiles, ht))) return rc;
  if ((rc = dev_upload(h, &h->d_conv_tiles, ct))) return rc;

  // ---- DP tables ---------------------------------------------------------------
  size_t act_cells = 0;
  for (int l = 0; l < n; ++l) if (h->lv[l].active) act_cells += (size_t)h->lv[l].cw * h->lv[l].ch;
  size_t allmaps = 0;
  for (const PartInfo& P : h->parts) if (P.p > 0) allmaps += P.K;
  std::vector<size_t> roundK(h->rounds.size(), 0);   // maps transformed in round r (per level)
  size_t maxK = 1;
  for (size_t r = 0; r < h->rounds.size(); ++r) {
    for (int fp : h->rounds[r]) roundK[r] += h->parts[fp].K;
    maxK = std::max(maxK, roundK[r]);
  }
  const bool fold = h->fold;
  // DT planes.  The passes' own pointer planes (int16) stay for the whole frame: back-tracking composes Ix / Iy from
  // them.  Score planes — fold: the x pass's output lives only until the round's y pass (one round's worth, reused
  // by every round), the y pass's output (the message source) keeps its own plane until the parent's x pass has
  // read it; legacy: both kept per map (a message may wait several rounds for a higher-indexed sibling), plus the
  // accumulated part scores.
  h->dt_cap_elems = std::max<size_t>(1, allmaps * act_cells);
  const size_t tmp_elems = fold ? std::max<size_t>(1, maxK * act_cells) : h->dt_cap_elems;
  h->d_dt_sdt = nullptr;
  if (!h->compact) {
    if ((rc = dev_alloc(h, &h->d_dt_tmpT, tmp_elems * ts))) return rc;
    if ((rc = dev_alloc(h, &h->d_dt_sdt, h->dt_cap_elems * ts))) return rc;
  }
  if ((rc = dev_alloc(h, &h->d_dt_ixT, h->dt_cap_elems))) return rc;
  if ((rc = dev_alloc(h, &h->d_dt_iy, h->dt_cap_elems))) return rc;
  if (!fold && (rc = dev_alloc(h, &h->d_acc, cells * h->nslots * ts))) return rc;
  if (h->split_parts) {
    // the features' bfloat16 parts (192 B per cell; binary16: 128 B) live from HOG to the end of the filter bank; the x pass's pointer planes from min()
    // to argmin(): the compact plan (whose stage buffers already refuse to be read once a later stage has reused them) puts both in
    // one region where the planes are large enough (person model: 300 B per cell); the default plan keeps them apart (pdf() may be
    // called again after min() there)
    const size_t split_elems = cells * h->split_parts * PBD_FLEN;
    if (h->compact && h->dt_cap_elems >= split_elems) h->d_feat_split = (uint16_t*)h->d_dt_ixT;
    else if ((rc = dev_alloc(h, &h->d_feat_split, split_elems))) return rc;
  }

  // DT LDS budget per block unless the longest line needs more at the minimum number of lines per block
  int maxlen = 1;
  for (int l = 0; l < n; ++l) if (h->lv[l].active) maxlen = std::max(maxlen, std::max(h->lv[l].cw, h->lv[l].ch));
  // block geometry, measured on MI355X (profiles/sweep_dt.sh).  float, lines with 16-bit links: two wavefronts and
  // 25 KB per block = 6 blocks = 3 wavefronts per SIMD (20 .. 40 KB swept); double (17 B per line element, an IEEE division
  // per intersection): one wavefront and 20 KB = 8 blocks per CU (0.93 ms against 1.28 with the float geometry)
  // Round 4 (profiles/experiments/README.md, seven frame sizes): while every line of the frame is short enough for byte links
  // (stride <= 256: 9 B per line element), float blocks of FOUR wavefronts and 40 KB — 4 blocks = 4 wavefronts per SIMD — beat
  // the two-wavefront / 25 KB blocks by 2-8 % of dp_min in batches (640x480: 0.328 -> 0.314 ms per frame, 0.600 -> 0.581 alone);
  // with 16-bit links (10 B per element: 1280x720, 1920x1080) they lose 9-12 %, and there the geometry above stays.
  const bool byte_links = dt_stride_for(maxlen) <= 256;
  // double (17 B per line element): two wavefronts and 40 KB per block — round 4, session 34: 0.473 / 0.721 ms per frame (batches / alone)
  // against 0.485 / 0.770 with one wavefront and 20 KB; round 5, session 3: 0.477 against 0.497 in batches, 893 against 879 frames/s
  // pbd_tune_plan: the other geometry measured on this handle's own frames (results are bit-identical under any geometry)
  const bool big_blocks = h->ts == 4 && h->dt_geom ? h->dt_geom == 1 : byte_links;
  h->dt_nt = h->ts == 8 ? 128 : (big_blocks ? 256 : PBD_DT_NT_DEFAULT);
  if (const char* e = PBD_PROBE_ENV("PBD_DT_NT")) h->dt_nt = std::max(64, std::min(256, atoi(e) & ~63));
  h->dt_nt_x = h->dt_nt;      // lanes of a fold x-pass block
  if (const char* e = PBD_PROBE_ENV("PBD_DT_NT_X")) h->dt_nt_x = std::max(64, std::min(256, atoi(e) & ~63));
  if (const char* e = PBD_PROBE_ENV("PBD_DT_SEG")) h->dt_seg = atoi(e);
  size_t dt_base = (h->ts == 8 ? 40 : (big_blocks ? 40 : 25)) * 1024;
  if (const char* e = PBD_PROBE_ENV("PBD_DT_BUDGET_KB")) dt_base = (size_t)atoi(e) * 1024;
  if (const char* e = PBD_PROBE_ENV("PBD_DT_BUDGET_B")) dt_base = (size_t)atoi(e);
  int max_mix = 4;
  if (fold) for (const PartInfo& P : h->parts) max_mix = std::max(max_mix, P.K);
  const size_t dt_need = dt_lds_bytes(dt_stride_for(maxlen), max_mix, h->ts, h->dt_nt);   // the longest line at the fewest lines a block can hold
  if (dt_need > 160 * 1024) return fail(h, PBD_ERR_UNSUPPORTED, "pyramid level too large for the LDS-resident distance transform");
  h->dt_lds = std::max(dt_base, dt_need);
  // A launch with fewer maps than the fullest round of the frame (the person tree: rounds of 4, 4, 4, 4, 2, 2, 2, 2, 1
  // parts) would leave LDS — and lanes — idle at the full budget, while every launch lasts as long as its blocks'
  // segments are: its blocks get proportionally FEWER lines (about the same number of blocks as the fullest launch),
  // i.e. more lanes per line and shorter segments.
  bool thin = false;
  size_t thin_min = 10 * 1024;
  if (const char* e = PBD_PROBE_ENV("PBD_DT_THIN")) thin = atoi(e) != 0;
  if (const char* e = PBD_PROBE_ENV("PBD_DT_THIN_MIN_KB")) thin_min = (size_t)atoi(e) * 1024;
  auto launch_budget = [&](size_t K_launch) {
    size_t bgt = h->dt_lds;
    if (thin && K_launch < maxK) bgt = std::max(thin_min, (size_t)((double)dt_base * (double)K_launch / (double)maxK));
    return std::min(std::max(bgt, dt_need), h->dt_lds);
  };
  std::vector<DtMap> maps;
  std::vector<DtTask> tasks;
  std::vector<FoldJob> folds;
  std::vector<unsigned long long> foldx;               // pbd_handle::d_foldx
  std::vector<ReduceJob> red;
  std::vector<ReduceBlock> redblk;
  h->rl.clear();
  std::vector<char> slot_init((size_t)h->nslots, 0);  // legacy: ncscores[fid].empty() emulation (same for every level)
  // plane offset of (part, level, mixture) in the per-map DT planes: parts in flat order, levels inside
  std::vector<size_t> part_scr(h->parts.size(), 0);
  { size_t o = 0; for (size_t fp = 0; fp < h->parts.size(); ++fp) if (h->parts[fp].p > 0) { part_scr[fp] = o; o += (size_t)h->parts[fp].K * act_cells; } }
  std::vector<size_t> lvl_scr(n, 0);  // prefix of active cells
  { size_t o = 0; for (int l = 0; l < n; ++l) { lvl_scr[l] = o; if (h->lv[l].active) o += (size_t)h->lv[l].cw * h->lv[l].ch; } }
  auto scr_of = [&](int fp, int l, int mm) {
    return part_scr[fp] + (size_t)h->parts[fp].K * lvl_scr[l] + (size_t)mm * h->lv[l].cw * h->lv[l].ch;
  };
  auto resp_plane = [&](int l, int fid) { return h->d_resp + (h->lv[l].cell_off * m.nfilters + (size_t)fid * h->lv[l].cw * h->lv[l].ch) * ts; };
  // compact: the transformed scores of (part, mixture) live in the mixture's own response plane
  auto sdt_plane = [&](int fp, int l, int mm) { return h->compact ? resp_plane(l, h->parts[fp].filterid[mm]) : h->d_dt_sdt + scr_of(fp, l, mm) * ts; };
  // children of every part, descending flat index (the order their messages are added in, src/DynamicProgram.cpp:95)
  std::vector<std::vector<int>> children(h->parts.size());
  for (int fp = (int)h->parts.size() - 1; fp >= 0; --fp)
    if (h->parts[fp].p > 0) children[h->part_offset[h->parts[fp].comp] + h->parts[fp].parent].push_back(fp);
  auto make_fold = [&](int fp, int l) {   // FoldJob of part fp at level l (its children's messages); -1 without children
    if (children[fp].empty()) return -1;
    const Level& L = h->lv[l];
    const size_t HW = (size_t)L.cw * L.ch;
    FoldJob J{};
    for (int c : children[fp]) {
      const PartInfo& C = h->parts[c];
      FoldChild& F = J.ch[J.nch++];
      F.K = C.K;
      for (int k = 0; k < PBD_FOLD_MAXMIX; ++k) F.sdt[k] = sdt_plane(c, l, std::min(k, C.K - 1));
      const int Lp = h->parts[fp].K;
      for (int k = 0; k < PBD_FOLD_MAXMIX; ++k)
        for (int mm = 0; mm < PBD_FOLD_MAXMIX; ++mm)
          F.bias[k][mm] = h->biasw[C.biasid[std::min(k, C.K - 1)] + std::min(mm, Lp - 1)];
      F.ok = h->d_pk + L.cell_off * h->nplanes + (size_t)C.plane0 * HW;
    }
    folds.push_back(J);
    return (int)folds.size() - 1;
  };
  // Workgroup b runs on XCD b % 8, each with its own L2.  A block writes its lines transposed, i.e. runs of a few
  // elements — a fraction of a 128-byte line; the neighbouring runs belong to the next tasks of the same map.
  // Order a launch's table so that `xcd_chunk` consecutive tasks share an XCD and the partial lines merge in one
  // L2 instead of going out to HBM from several.
  int xchunk = h->xcd_chunk;
  if (const char* e = PBD_PROBE_ENV("PBD_DT_XCD_CHUNK")) xchunk = atoi(e);
  auto xcd_order = [xchunk](std::vector<DtTask>& v) {
    const int c = xchunk;
    if (c <= 0) return;
    const size_t win = (size_t)8 * c, full = v.size() / win * win;
    std::vector<DtTask> o(v);
    for (size_t b = 0; b < full; ++b) {
      const size_t xcd = b & 7, idx = b >> 3;
      o[b] = v[((idx / c) * 8 + xcd) * c + idx % c];
    }
    v.swap(o);
  };
  auto launch_lds = [&](const std::vector<DtTask>& v, int nt) {
    size_t lds = 0;
    for (const DtTask& t : v) lds = std::max(lds, dt_lds_bytes(t.g.stride, t.g.lpb, h->ts, nt));
    if (const char* e = PBD_PROBE_ENV("PBD_DT_LDS_REQUEST_KB")) lds = std::max(lds, (size_t)atoi(e) * 1024);   // occupancy probe
    return lds;
  };
  const bool dbg_plan = PBD_PROBE_ENV("PBD_DEBUG_PLAN") != nullptr;
  // A launch whose blocks do not all fit on the chip at once lasts two block times instead of one (measured: the
  // fold x pass of the 4-part rounds, 1772 blocks for 1536 slots at 25 KB: 100 us instead of 45).  Blocks that hold
  // whole rows of all mixtures quantise badly (12 lines where 15 would fit), so such a launch gets the smallest
  // larger budget at which its blocks are resident together (fewer, larger blocks; fewer blocks per CU).
  int ncu = 256;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, h->opt.device) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount; }
  h->ncu = ncu;
  auto count_blocks = [&](const std::vector<int>& rnd, size_t budget, bool fold_x, bool ypass, bool round_lanes, size_t* lds_out) {
    size_t nb = 0, lds = 0;
    for (int l = 0; l < n; ++l) {
      const Level& L = h->lv[l];
      if (!L.active || L.cw == 0 || L.ch == 0) continue;
      const int len = ypass ? L.ch : L.cw, nlines = ypass ? L.cw : L.ch;
      if (fold_x) {
        for (int fp : rnd) {
          const DtGroup g = dt_group(0, h->parts[fp].K, nlines, len, budget, h->ts, h->dt_nt, h->dt_seg, 0);
          nb += (size_t)(nlines + g.lpb / g.nmaps - 1) / (g.lpb / g.nmaps);
          lds = std::max(lds, dt_lds_bytes(g.stride, g.lpb, h->ts, h->dt_nt));
        }
      } else {
        int nm = 0;
        for (int fp : rnd) nm += h->parts[fp].K;
        const DtGroup g = dt_group(0, nm, nlines, len, budget, h->ts, h->dt_nt, h->dt_seg, -1, round_lanes);
        nb += ((size_t)nm * nlines + g.lpb - 1) / g.lpb;
        lds = std::max(lds, dt_lds_bytes(g.stride, g.lpb, h->ts, h->dt_nt));
      }
    }
    *lds_out = lds;
    return nb;
  };
  // Launch geometry: the base budget with full-lane lines per block if all blocks of the launch are then resident at
  // once; else the same without the rounding; else the smallest larger budget that makes them resident (fewer, larger
  // blocks; up to 1.6 x); else the base.  (Measured: budgets BELOW the base — more, shorter blocks, 8 per CU — are
  // slower: dp_min 0.79 / 0.82 ms fold / three-kernel against 0.72 / 0.76, more contention per CU.)
  struct Geo { size_t budget; bool round; };
  auto launch_geometry = [&](const std::vector<int>& rnd, size_t base, bool fold_x, bool ypass) {
    const Geo dflt{base, true};
    // fold launches keep the base budget: their blocks hold whole rows of all mixtures and quantise badly (1772 blocks
    // for 1536 slots in the 4-part rounds), but the budgets that make them resident at once (36-40 KB, 4 per CU) cost
    // more with four frames in flight than the second wave of blocks does (measured: 1 170 frames/s and dp_min 0.70 ms
    // at 40 KB, 1 181 / 0.73 at 28 KB, 1 235 / 0.725 at the base 25 KB)
    if (fold_x || PBD_PROBE_ENV("PBD_DT_NO_RESIDENT")) return dflt;
    const int waves_blk = std::max(1, h->dt_nt / 64);
    auto resident = [&](size_t b, bool rnd_lanes) {
      size_t lds = 0;
      const size_t nb = count_blocks(rnd, b, fold_x, ypass, rnd_lanes, &lds);
      const size_t per_cu = std::min<size_t>(160 * 1024 / std::max<size_t>(lds, 1), 24 / waves_blk);
      return nb <= per_cu * ncu;
    };
    if (resident(base, true)) return dflt;
    if (!fold_x && resident(base, false)) return Geo{base, false};
    for (size_t b = base + 1024; b <= base * 8 / 5 && b <= 150 * 1024; b += 1024) {
      if (resident(b, true)) return Geo{b, true};
      if (!fold_x && resident(b, false)) return Geo{b, false};
    }
    return dflt;
  };
  for (size_t r = 0; r < h->rounds.size(); ++r) {
    const std::vector<int>& rnd = h->rounds[r];
    pbd_handle::RoundLaunch R{};
    const bool fold_x = fold && r > 0;   // round 0 = the leaves: their lines are their raw responses
    if (rnd.empty()) { h->rl.push_back(R); continue; }
    size_t budget_x = launch_budget(roundK[r]), budget = budget_x;
    Geo geox = launch_geometry(rnd, budget_x, fold_x, false), geoy = launch_geometry(rnd, budget, false, true);
    if (const char* e = PBD_PROBE_ENV("PBD_DT_BUDGET_X_KB")) { if (fold_x) geox = Geo{std::max(dt_need, (size_t)atoi(e) * 1024), true}; }
    budget_x = geox.budget; budget = geoy.budget;
    std::vector<DtTask> xt, yt;
    for (int l = 0; l < n && !rnd.empty(); ++l) {
      const Level& L = h->lv[l];
      if (!L.active || L.cw == 0 || L.ch == 0) continue;
      const size_t HW = (size_t)L.cw * L.ch;
      const int gx_map0 = (int)maps.size();
      int gx_nmaps = 0;
      std::vector<DtMap> ymaps;
      size_t tmp_round = 0;   // fold: maps of this level in front of the part's, in the round's x-pass output (level-major: all maps of a
                              // level back to back, so that every block of the y pass reads ONE contiguous run — DtTask::src0)
      for (int fp : rnd) {
        const PartInfo& P = h->parts[fp];
        const int part_map0 = (int)maps.size();
        for (int mm = 0; mm < P.K; ++mm) {
          const int fid = P.filterid[mm], did = P.defid[mm];
          const size_t so = scr_of(fp, l, mm);
          const size_t to = fold ? roundK[r] * lvl_scr[l] + (tmp_round + (size_t)mm) * HW : so;
          const char* src = (!fold && slot_init[P.slot[mm]]) ? h->d_acc + (L.cell_off * h->nslots + (size_t)P.slot[mm] * HW) * ts
                                                            : resp_plane(l, fid);
          const float* wv = &h->defw[(size_t)did * 4];
          maps.push_back(dt_map(src, h->d_dt_tmpT + to * ts, h->d_dt_ixT + so, wv[0], wv[1], h->anchors[did * 2], 1));
          ymaps.push_back(dt_map(h->d_dt_tmpT + to * ts, sdt_plane(fp, l, mm), h->d_dt_iy + so, wv[2], wv[3], h->anchors[did * 2 + 1], 0));
          gx_nmaps++;
        }
        tmp_round += (size_t)P.K;
        if (fold_x) dt_add_tasks(dt_group(part_map0, P.K, L.ch, L.cw, budget_x, h->ts, h->dt_nt_x, h->dt_seg, make_fold(fp, l)), xt);
      }
      if (!fold_x) dt_add_tasks(dt_group(gx_map0, gx_nmaps, L.ch, L.cw, budget_x, h->ts, h->dt_nt, h->dt_seg, -1, geox.round), xt, &maps, h->ts);
      const DtGroup gy = dt_group((int)maps.size(), gx_nmaps, L.cw, L.ch, budget, h->ts, h->dt_nt, h->dt_seg, -1, geoy.round);
      for (auto& my : ymaps) maps.push_back(my);
      dt_add_tasks(gy, yt, &maps, h->ts);
      if (dbg_plan)
        fprintf(stderr, "plan: round %zu level %d  x: len %d lines %d maps %d lpb %d  y: len %d lines %d lpb %d  (budget %zu)\n", r, l,
                L.cw, L.ch, gx_nmaps, xt.empty() ? 0 : xt.back().g.lpb, gy.len, gy.nlines, gy.lpb, budget);
    }
    if (const char* e = PBD_PROBE_ENV("PBD_DEBUG_DUP")) {   // scaling probe: every DT block issued n times (identical outputs)
      const int ndup = atoi(e);
      const std::vector<DtTask> x0 = xt, y0 = yt;
      for (int i = 1; i < ndup; ++i) { xt.insert(xt.end(), x0.begin(), x0.end()); yt.insert(yt.end(), y0.begin(), y0.end()); }
    }
    if (PBD_PROBE_ENV("PBD_DT_REVERSE")) { std::reverse(xt.begin(), xt.end()); std::reverse(yt.begin(), yt.end()); }   // probe: coarse levels' blocks first
    dt_mark_fused(xt, maps.data(), h->ts);
    dt_mark_fused(yt, maps.data(), h->ts);
    xcd_order(xt);
    xcd_order(yt);
    R.lds_x = launch_lds(xt, fold_x ? h->dt_nt_x : h->dt_nt); R.lds_y = launch_lds(yt, h->dt_nt); R.fold_x = fold_x ? 1 : 0;
    if (dbg_plan) fprintf(stderr, "plan: round %zu: %zu x blocks (%zu B LDS%s), %zu y blocks (%zu B)\n", r, xt.size(), R.lds_x, fold_x ? ", fold" : "", yt.size(), R.lds_y);
    if (fold_x) {   // the loader's first addresses of every fold x task, in task order (pbd_handle::d_foldx)
      R.foldx0 = foldx.size();
      for (const DtTask& t : xt) {
        const DtGroup& g = t.g;
        for (int mm = 0; mm < 8; ++mm) foldx.push_back((unsigned long long)(uintptr_t)maps[(size_t)g.map0 + std::min(mm, g.nmaps - 1)].src);
        const unsigned long long* cq = (const unsigned long long*)&folds[(size_t)g.fold].ch[0];   // sdt[8], ok (k_dp.hip: fold_child_qw)
        for (int i = 0; i < 9; ++i) foldx.push_back(cq[i]);
        foldx.push_back((unsigned long long)folds[(size_t)g.fold].nch);
      }
    }
    R.xtask0 = (int)tasks.size(); R.nxtasks = (int)xt.size();
    tasks.insert(tasks.end(), xt.begin(), xt.end());
    R.ytask0 = (int)tasks.size(); R.nytasks = (int)yt.size();
    tasks.insert(tasks.end(), yt.begin(), yt.end());
    // legacy: reduce waves of this round (slot state is advanced once per wave)
    for (size_t wi = 0; !fold && wi < h->red_rounds[r].size(); ++wi) {
      const std::vector<int>& wave = h->red_rounds[r][wi];
      pbd_handle::ReduceWave Wv{(int)redblk.size(), 0};
      std::vector<int> parents;  // distinct parents, in first-appearance order
      for (int fp : wave) {
        const int pf = h->part_offset[h->parts[fp].comp] + h->parts[fp].parent;
        if (std::find(parents.begin(), parents.end(), pf) == parents.end()) parents.push_back(pf);
      }
      for (int l = 0; l < n; ++l) {
        const Level& L = h->lv[l];
        if (!L.active || L.cw == 0 || L.ch == 0) continue;
        const size_t HW = (size_t)L.cw * L.ch;
        for (int pf : parents) {
          const PartInfo& Par = h->parts[pf];
          ReduceJob J{};
          J.H = L.ch; J.W = L.cw; J.L = Par.K;
          for (int pm = 0; pm < Par.K; ++pm) {
            char* accp = h->d_acc + (L.cell_off * h->nslots + (size_t)Par.slot[pm] * HW) * ts;
            J.par_in[pm] = slot_init[Par.slot[pm]] ? accp : resp_plane(l, Par.filterid[pm]);
            J.par_out[pm] = accp;
          }
          for (int fp : wave) {  // `wave` is in descending child order
            const PartInfo& P = h->parts[fp];
            if (h->part_offset[P.comp] + P.parent != pf) continue;
            ReduceChild& C = J.ch[J.nch++];
            C.sdt = sdt_plane(fp, l, 0);
            C.ok = h->d_pk + L.cell_off * h->nplanes + (size_t)P.plane0 * HW;
            C.K = P.K;
            for (int mm = 0; mm < P.K; ++mm) C.bias_off[mm] = P.biasid[mm];
          }
          for (unsigned c0 = 0; c0 < (unsigned)HW; c0 += 256) redblk.push_back(ReduceBlock{(int)red.size(), c0});
          red.push_back(J);
        }
      }
      for (int pf : parents)
        for (int pm = 0; pm < h->parts[pf].K; ++pm) slot_init[h->parts[pf].slot[pm]] = 1;
      Wv.nblks = (int)redblk.size() - Wv.blk0;
      R.waves.push_back(Wv);
    }
    h->rl.push_back(R);
  }

  // root jobs + backtracking info
  std::vector<RootJob> rj;
  std::vector<ReduceBlock> rootblk;   // k_root's blocks: (job, first cell)
  std::vector<BackLevel> bl((size_t)n * m.ncomponents);
  unsigned rcells = 0;
  for (int l = 0; l < n; ++l) {
    const Level& L = h->lv[l];
    const size_t HW = (size_t)L.cw * L.ch;
    for (int c = 0; c < m.ncomponents; ++c) {
      BackLevel& B = bl[(size_t)l * m.ncomponents + c];
      const size_t po = L.cell_off * h->nplanes + (size_t)h->comp_plane0[c] * HW;
      B.pk = h->d_pk + po;
      B.rootv = h->d_rootv + (L.cell_off * m.ncomponents + (size_t)c * HW) * ts;
      B.rooti = h->d_rooti + L.cell_off * m.ncomponents + (size_t)c * HW;
      B.H = L.ch; B.W = L.cw; B.scale = L.scale;
      if (!L.active || HW == 0) continue;
      const PartInfo& R0 = h->parts[h->part_offset[c]];
      RootJob J{};
      for (int kk = 0; kk < PBD_MAX_MIX; ++kk) {   // entries beyond K repeat mixture K - 1 (the fold's register arrays are never predicated)
        const int k = std::min(kk, R0.K - 1);
        J.score[kk] = (!fold && slot_init[R0.slot[k]]) ? h->d_acc + (L.cell_off * h->nslots + (size_t)R0.slot[k] * HW) * ts
                                                       : resp_plane(l, R0.filterid[k]);
      }
      J.rootv = (void*)B.rootv; J.rooti = (int*)B.rooti;
      J.H = L.ch; J.W = L.cw; J.K = R0.K; J.level = l; J.comp = c;
      J.bias = h->biasw[R0.biasid[0]];  // root.bias(0)[0], DynamicProgram.cpp:165
      J.cell0 = rcells;
      J.fold = fold ? make_fold(h->part_offset[c], l) : -1;   // fold: the root's messages are folded by k_root
      rcells += (unsigned)HW;
      for (unsigned c0 = 0; c0 < (unsigned)HW; c0 += 256) rootblk.push_back(ReduceBlock{(int)rj.size(), c0});
      rj.push_back(J);
    }
  }
  if ((rc = dev_upload(h, &h->d_dtmaps, maps))) return rc;
  if ((rc = dev_upload(h, &h->d_dttasks, tasks))) return rc;
  if ((rc = dev_upload(h, &h->d_foldjobs, folds))) return rc;
  if ((rc = dev_upload(h, &h->d_foldx, foldx))) return rc;
  if ((rc = dev_upload(h, &h->d_redjobs, red))) return rc;
  if ((rc = dev_upload(h, &h->d_redblocks, redblk))) return rc;
  h->n_rootjobs = (int)rj.size();
  h->root_cells = rcells;
  h->root_maxcells = 0;
  for (const RootJob& J : rj) h->root_maxcells = std::max(h->root_maxcells, (unsigned)J.H * (unsigned)J.W);
  if ((rc = dev_upload(h, &h->d_rootjobs, rj))) return rc;
  h->n_rootblocks = (int)rootblk.size();
  if ((rc = dev_upload(h, &h->d_rootblocks, rootblk))) return rc;
  if ((rc = dev_upload(h, &h->d_back, bl))) return rc;
  // where the DT pointer planes of (level, part) live: back-tracking composes Ix / Iy from them on the fly
  h->scr_base.assign((size_t)n * h->parts.size(), 0);
  for (int l = 0; l < n; ++l)
    for (size_t fp = 0; fp < h->parts.size(); ++fp)
      if (h->parts[fp].p > 0 && h->lv[l].active) h->scr_base[(size_t)l * h->parts.size() + fp] = scr_of((int)fp, l, 0);
  if ((rc = dev_upload(h, &h->d_scr_base, h->scr_base))) return rc;
  h->fw = w; h->fh = hgt; h->fcn = cn; h->fdepth = depth; h->fesz = esz;
  return PBD_OK;
}

// ---------------------------------------------------------------------------
// stages
// ---------------------------------------------------------------------------
static int run_image_pyramid(pbd_handle* h, const uint8_t* d_src, int stride) {
  // one launch for the first octave of every frame of the batch (cv::resize), one per octave step below it (cv::pyrDown)
  for (size_t i = 0; i < h->pyr_launches.size(); ++i) {
    const pbd_handle::PyrLaunch& P = h->pyr_launches[i];
    if (h->fdepth != PBD_DEPTH_8U) {   // 16-bit / float / double frames (pbd_detect_image): the plain per-element kernels
      if (i == 0) launch_resize_any(h->d_pyrjobs + P.job0, P.njobs, P.maxpix, h->fcn, h->fdepth, stride, d_src, h->d_pyr, h->stream);
      else launch_pyrdown_any(h->d_pyrjobs + P.job0, P.njobs, P.maxpix, h->fcn, h->fdepth, h->d_pyr, h->stream);
    } else
    if (i == 0) launch_resize(h->d_pyrjobs + P.job0, P.njobs, P.maxpix, h->fcn, stride, d_src, h->d_pyr, h->stream);
    else launch_pyrdown(h->d_pyrjobs + P.job0, P.njobs, P.maxw, P.maxh, h->fcn, h->d_pyr, h->stream);
  }
  LAUNCHCHK(h, "image pyramid");
  h->have_pyr = true;
  if (h->compact) h->have_dp = h->min_ran = false;   // the level images + features share their memory with the Ik planes / the x pass's scratch:
                                                     // the previous min()'s tables are gone (a plane handed in later is then NOT on top of a min())
  return PBD_OK;
}

static int run_hog(pbd_handle* h) {
  uint16_t* split = h->split_parts ? h->d_feat_split : nullptr;
  launch_hog(h->d_hog_tiles, h->n_hog_tiles, h->d_levels, h->d_pyr, h->d_feat, h->ts, h->fcn, h->md.sbin, h->hog_tc, h->d_hog_lut, split, h->split_parts, h->fdepth, h->stream);
  LAUNCHCHK(h, "HOG");
  h->feat_split_ok = split != nullptr;
  h->have_feat = true;
  compact_mark_feat(h, true);
  if (h->compact) h->have_dp = h->min_ran = false;
  return PBD_OK;
}

static int run_pdf(pbd_handle* h) {
  const pbd_model_desc& m = h->md;
  if (h->conv_mode == PBD_CONV_SPLIT_F16) {
    if (!h->feat_split_ok) {
      launch_feat_split16((const float*)h->d_feat, h->d_feat_split, h->cells, h->stream);
      h->feat_split_ok = true;
    }
    static const int svariant16 = PBD_PROBE_ENV("PBD_SPLIT_VARIANT") ? atoi(PBD_PROBE_ENV("PBD_SPLIT_VARIANT")) : 0;   // tuning builds
    launch_conv_split16(h->d_conv_tiles, h->n_conv_tiles, h->d_levels, h->d_feat_split, h->d_wS, (float*)h->d_resp, m.nfilters, m.kh, m.kw, h->d_split_oscale, svariant16, h->stream);
  } else if (h->conv_mode == PBD_CONV_SPLIT) {
    // the features' three exact bfloat16 parts: written by k_hog's epilogue; features handed in by the caller (pbd_set_level_features)
    // are split here (a pass over 25 MB per frame).  Then the bank on the bf16 matrix units
    if (!h->feat_split_ok) {
      launch_feat_split((const float*)h->d_feat, h->d_feat_split, h->cells, h->stream);
      h->feat_split_ok = true;
    }
    static const int svariant = PBD_PROBE_ENV("PBD_SPLIT_VARIANT") ? atoi(PBD_PROBE_ENV("PBD_SPLIT_VARIANT")) : 0;   // tuning builds
    if (svariant == 6 && m.kh == 5 && m.kw == 5)
      launch_conv_split_persistent(h->d_conv_tiles, h->n_conv_tiles, h->d_levels, h->d_feat_split, h->d_wS, (float*)h->d_resp, m.nfilters, h->ncu, h->stream);
    else
    launch_conv_split(h->d_conv_tiles, h->n_conv_tiles, h->d_levels, h->d_feat_split, h->d_wS, (float*)h->d_resp, m.nfilters, m.kh, m.kw, svariant, h->stream);
  } else if (h->conv_mode == PBD_CONV_MFMA)
    if (h->ts == 8) launch_conv_mfma_f64(h->d_conv_tiles, h->n_conv_tiles, h->d_levels, (const double*)h->d_feat, (const double*)h->d_wT,
                                         (const double*)h->d_wT + (size_t)m.kh * m.kw * m.flen * h->nfpad + m.flen, (double*)h->d_resp, m.nfilters, h->nfpad, m.kh, m.kw, h->stream);
    else {
      // default (20): 16x16x4 MFMA, tile staged in two channel halves, TWO 16-filter n-tiles per workgroup, B operand by 16-byte
      // loads from the [tap][half][k][n][u] copy of the filters (k_conv_mfma16<float, 2, 3, 2, true>: 27 KB of LDS per workgroup, so
      // DT blocks of other frames co-reside on the CU).  Two n-tiles per workgroup: alone the same time as one, but every tile is
      // staged half as often and with frames in flight that VALU / LDS time goes to the other frames' DT blocks (1 392 vs 1 331
      // frames/s, batches of 4 on 3 handles).  16-byte B loads: 0.339 vs 0.388 ms sequential, 1 419 vs 1 391 frames/s (eight
      // global_load_dword per 32 MFMAs cost the MFMA pipe a quarter of its issue rate: tests/tools/mfma_rate_probe.hip).
      // PBD_MFMA_VARIANT (probe / tuning builds): 0 = the older 32x32x2 kernel, 1 = whole tile, 2 = halves at 5 waves/SIMD, 3 = one
      // n-tile, 4 = channel quarters, 5-9 = n-tile counts with 4-byte B loads, 10 / 11 / 19 = the persistent double-buffered
      // kernel k_conv_glds at 2 / 3 / 1 workgroups per CU (0.354 ms sequential, 1 353-1 378 frames/s), 21 / 22 = one n-tile /
      // 2 waves per SIMD register allocation with 16-byte B loads
      static const int lds_req = g_conv_lds_req_kb = PBD_PROBE_ENV("PBD_CONV_LDS_KB") ? atoi(PBD_PROBE_ENV("PBD_CONV_LDS_KB")) : 0;
      (void)lds_req;
      static const int variant = PBD_PROBE_ENV("PBD_MFMA_VARIANT") ? atoi(PBD_PROBE_ENV("PBD_MFMA_VARIANT")) : 20;
      if (variant >= 10 && variant < 20 && m.kh == 5 && m.kw == 5 && m.flen == PBD_FLEN)
        launch_conv_glds_f32(h->d_conv_tiles, h->n_conv_tiles, h->d_levels, (const float*)h->d_feat,
                             (const float*)h->d_wT + (size_t)m.kh * m.kw * m.flen * h->nfpad + m.flen /* [tap][half][k][n][s] copy */, (float*)h->d_resp, m.nfilters, h->nfpad,
                             (const float*)h->d_wT + (size_t)m.kh * m.kw * m.flen * h->nfpad /* border cell */, variant == 19 ? 1 : variant == 18 ? 0 : variant - 8, h->ncu, h->stream);
      else if (variant || m.kh != 5 || m.kw != 5)   // (filters other than 5 x 5: the same kernel with a run-time tap loop)
        launch_conv_mfma16_f32(h->d_conv_tiles, h->n_conv_tiles, h->d_levels, (const float*)h->d_feat, (const float*)h->d_wT,
                               (const float*)h->d_wT + 2 * (size_t)m.kh * m.kw * m.flen * h->nfpad + m.flen, (float*)h->d_resp, m.nfilters, h->nfpad, variant, h->stream, m.kh, m.kw);
      else
        launch_conv_mfma(h->d_conv_tiles, h->n_conv_tiles, h->d_levels, (const float*)h->d_feat, (const float*)h->d_wT, (float*)h->d_resp, m.nfilters, h->nfpad, m.kh, m.kw, h->stream);
    }
  else
    launch_conv_exact(h->d_conv_tiles, h->n_conv_tiles, h->d_levels, h->d_feat, h->d_wT, h->d_resp, h->ts, m.nfilters, h->nfpad, m.kh, m.kw, h->stream);
  LAUNCHCHK(h, "filter bank");
  h->have_resp = true;
  compact_mark_resp(h, true);
  return PBD_OK;
}

// pbd_options.reserved[0] = sz > 0: nonMaximaSuppression(rootv, sz) of every (level, component) plane on the device, then the hits
static void nms_and_rescan(pbd_handle* h) {
  hipMemsetAsync(h->d_nms_mask, 0, h->cells * (size_t)h->md.ncomponents, h->stream);
  launch_nms_roots(h->d_rootjobs, h->n_rootjobs, h->root_maxcells, h->d_rootv, h->ts, h->nms_sz, h->d_nms_mask, h->stream);
  launch_root(h->d_rootjobs, h->d_rootblocks, h->n_rootblocks, (double)h->md.thresh, h->d_cand_count, h->d_cand_rec,
              h->opt.max_candidates, h->ts, h->d_foldjobs, h->d_biasw, 1, h->fold_mix, h->d_nms_mask, h->d_rootv, h->stream);
}

static int run_dp_min(pbd_handle* h) {
  const bool dpt = h->profiling && h->dp_timer_on;
  if (dpt) hipEventRecord(h->ev_dp0, h->stream);
  // one chain of rounds on the handle's stream.  fold: x pass (from round 1 on its loader folds the children's
  // messages) + y pass per round, the root's messages folded by k_root: 2 * rounds + 1 launches; legacy (models that
  // alias a filter id inside a component, or more than 8 mixtures): + the round's reduce launches.
  for (auto& R : h->rl) {
    launch_dt_pass(h->d_dttasks + R.xtask0, R.nxtasks, h->d_dtmaps, R.fold_x ? h->d_foldjobs : nullptr, R.fold_x ? h->d_foldx + R.foldx0 : nullptr, h->d_biasw, R.lds_x, h->ts, R.fold_x ? h->dt_nt_x : h->dt_nt, h->fold_mix, h->stream);
    launch_dt_pass(h->d_dttasks + R.ytask0, R.nytasks, h->d_dtmaps, nullptr, nullptr, h->d_biasw, R.lds_y, h->ts, h->dt_nt, 0, h->stream);
    for (auto& Wv : R.waves)
      launch_reduce(h->d_redjobs, h->d_redblocks + Wv.blk0, Wv.nblks, h->d_biasw, h->opt.dt_correct_ptr, h->ts, h->stream);
  }
  hipMemsetAsync(h->d_cand_count, 0, sizeof(int), h->stream);
  if (h->nms_sz > 0) {
    // score-map NMS in front of the back-tracking (src/nms.cpp:84-129; the reference's call site is commented out,
    // src/PartsBasedDetector.cpp:86): the root tables are written with no hit (threshold +inf), the block NMS runs on the
    // resident rootv planes, and the hit list is then compacted from (score > thresh) AND (local maximum)
    launch_root(h->d_rootjobs, h->d_rootblocks, h->n_rootblocks, INFINITY, h->d_cand_count, h->d_cand_rec,
                h->opt.max_candidates, h->ts, h->d_foldjobs, h->d_biasw, 0, h->fold_mix, nullptr, nullptr, h->stream);
    nms_and_rescan(h);
  } else
  launch_root(h->d_rootjobs, h->d_rootblocks, h->n_rootblocks, (double)h->md.thresh, h->d_cand_count, h->d_cand_rec,
              h->opt.max_candidates, h->ts, h->d_foldjobs, h->d_biasw, 0, h->fold_mix, nullptr, nullptr, h->stream);
  if (dpt) hipEventRecord(h->ev_dp1, h->stream);
  h->dp_timed = dpt;
  LAUNCHCHK(h, "DP min");
  h->have_dp = true;
  h->min_ran = true;
  h->ext_ptr = false;   // back-tracking reads this min()'s own tables again
  h->root_dirty = false;
  if (h->compact) {     // their memory now holds the DP's planes: every feature / response plane is stale until produced or handed in again
    h->have_pyr = h->have_feat = h->have_resp = false;
    h->feat_split_ok = false;   // (the split bank's copy of the features shared the x pass's pointer planes)
    compact_mark_feat(h, false);
    compact_mark_resp(h, false);
  }
  return PBD_OK;
}

static int run_argmin_enqueue(pbd_handle* h) {
  if (h->root_dirty) {   // root tables injected since min(): the hits are those of the tables now on the device
    hipMemsetAsync(h->d_cand_count, 0, sizeof(int), h->stream);
    if (h->nms_sz > 0) nms_and_rescan(h);
    else
    launch_root(h->d_rootjobs, h->d_rootblocks, h->n_rootblocks, (double)h->md.thresh, h->d_cand_count, h->d_cand_rec,
                h->opt.max_candidates, h->ts, h->d_foldjobs, h->d_biasw, 1, h->fold_mix, nullptr, nullptr, h->stream);
    h->root_dirty = false;
  }
  // Round 6: a handle that is not a member of an RCCL-gathering group lets the back-tracking kernel write the records and the count straight
  // into its pinned host buffers (hipHostMalloc: device-mapped, coherent): no copy nodes behind the kernel, and never a second copy for
  // records beyond a first block.  Group members keep the device buffer: the all-gather reads it.
  const bool zero_copy = PBD_ARGMIN_ZERO_COPY && !h->d_gsend;
  launch_backtrack(h->d_cand_count, h->d_cand_rec, h->opt.max_candidates, h->d_back, h->md.ncomponents, h->d_parent,
                   h->d_plane0, h->d_nparts, h->max_parts, h->md.kh, zero_copy ? h->h_cand_out : h->d_cand_out, h->cand_stride, h->ts, h->d_flat,
                   h->d_depth, h->max_depth, (int)h->parts.size(), h->d_scr_base, h->d_dt_ixT, h->d_dt_iy,
                   h->opt.dt_correct_ptr, h->ext_ptr ? h->d_extx : nullptr, h->d_exty, h->d_ext_base, zero_copy ? h->h_cand_count : nullptr, h->stream);
  LAUNCHCHK(h, "argmin");
  if (zero_copy) { h->pending = true; h->out_on_host = true; return PBD_OK; }
  h->out_on_host = false;
  // members of an RCCL-gathering group send a fixed block (pbd_group.cpp sizes its buffers for PBD_FIRST_COPY records)
  if (h->d_gsend || h->first_copy < kFirstCopy * h->batch) h->first_copy = kFirstCopy * h->batch;
  const int first = std::min(h->first_copy, h->opt.max_candidates);
  if (h->d_gsend) {   // member of an RCCL-gathering pbd_group: pack {count, first records} for the all-gather instead of the D2H
    HIPCHK(h, hipMemcpyAsync(h->d_gsend, h->d_cand_count, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_gsend + 16, h->d_cand_out, h->cand_stride * first, hipMemcpyDeviceToDevice, h->stream));
  } else {
    HIPCHK(h, hipMemcpyAsync(h->h_cand_count, h->d_cand_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_cand_out, h->d_cand_out, h->cand_stride * first, hipMemcpyDeviceToHost, h->stream));
  }
  h->pending = true;
  return PBD_OK;
}

// frame finished on the stream: timers, and the records beyond the first block (rare) fetched into h_cand_out.
// `found` = the device-side count (h_cand_count[0], or the count a group gather delivered).
int pbd_i_finish_frame(pbd_handle* h, int found) {
  h->pending = false;
  if (h->dp_timed && h->have_dp) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev_dp0, h->ev_dp1) == hipSuccess) { h->dp_ms_sum += ms; h->dp_frames++; }
  }
  const int n = std::min(found, h->opt.max_candidates);
  const int first = h->out_on_host ? n : std::min(h->first_copy, h->opt.max_candidates);   // (zero-copy: every record is on the host already)
  if (n > first) {
    HIPCHK(h, hipMemcpyAsync(h->h_cand_out + h->cand_stride * first, h->d_cand_out + h->cand_stride * first,
                             h->cand_stride * (n - first), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!h->d_gsend) {
      // this geometry / threshold yields more records than the first copy carries: size it for what was seen (+25 %), so
      // that the following frames need no second copy and no second synchronisation.  The copy's size is baked into a
      // captured graph: drop it, the next frame captures again (host cost of one capture, once).
      h->first_copy = std::min(h->opt.max_candidates, n + n / 4 + 16);
      if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    }
  }
  if (found > h->opt.max_candidates) return fail(h, PBD_ERR_CAPACITY, "device candidate capacity exceeded; raise pbd_options.max_candidates");
  return PBD_OK;
}

// Candidate records (cand_stride bytes each, possibly from several handles of one group: recs[i] points at record i)
// -> the caller's arrays, ordered like a single-threaded reference run: level, component, row-major root location.
int pbd_i_emit(pbd_handle* h, const std::vector<const char*>& recs, pbd_candidate_head* heads, int32_t* boxes,
               int32_t* locs, int capacity) {
  const int mp = h->max_parts, n = (int)recs.size();
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  auto key = [&](int i, int k) -> int {
    const char* o = recs[i];
    const pbd_candidate_head* hd = (const pbd_candidate_head*)o;
    const int32_t* lc = (const int32_t*)(o + sizeof(pbd_candidate_head)) + (size_t)mp * 4;
    return k == 0 ? hd->level : k == 1 ? hd->component : k == 2 ? lc[1] : lc[0];
  };
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    for (int k = 0; k < 4; ++k) { int ka = key(a, k), kb = key(b, k); if (ka != kb) return ka < kb; }
    return false;
  });
  if (n > capacity) return fail(h, PBD_ERR_CAPACITY, "output capacity too small");
  for (int i = 0; i < n; ++i) {
    const char* o = recs[order[i]];
    if (heads) heads[i] = *(const pbd_candidate_head*)o;
    const int32_t* b = (const int32_t*)(o + sizeof(pbd_candidate_head));
    if (boxes) memcpy(boxes + (size_t)i * mp * 4, b, sizeof(int32_t) * mp * 4);
    if (locs) memcpy(locs + (size_t)i * mp * 3, b + (size_t)mp * 4, sizeof(int32_t) * mp * 3);
  }
  return PBD_OK;
}

static int collect(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* count) {
  if (!h->pending) return fail(h, PBD_ERR_STATE, "collect without a pending detect");
  if (h->d_gsend) return fail(h, PBD_ERR_STATE, "handle belongs to an RCCL-gathering pbd_group: collect through the group");
  if (h->batch > 1) return fail(h, PBD_ERR_STATE, "a batch of frames is pending: collect it with pbd_detect_batch_collect");
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const int found = h->h_cand_count[0];
  if (count) *count = found;
  int rc = pbd_i_finish_frame(h, found);
  if (rc) return rc;
  std::vector<const char*> recs((size_t)found);
  for (int i = 0; i < found; ++i) recs[i] = h->h_cand_out + h->cand_stride * i;
  return pbd_i_emit(h, recs, heads, boxes, locs, capacity);
}
int pbd_i_collect(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* count) {
  int rc = collect(h, heads, boxes, locs, capacity, count);
  return rc;
}

static int enqueue_all(pbd_handle* h, const uint8_t* d_src, int stride);
int pbd_i_enqueue_all(pbd_handle* h, const uint8_t* d_src, int stride) { return enqueue_all(h, d_src, stride); }
static int enqueue_stages(pbd_handle* h, const uint8_t* d_src, int stride) {
  int rc;
  const bool prof = h->profiling;
  if (prof) hipEventRecord(h->ev[0], h->stream);
  if ((rc = run_image_pyramid(h, d_src, stride))) return rc;
  if (prof) hipEventRecord(h->ev[1], h->stream);
  if ((rc = run_hog(h))) return rc;
  if (prof) hipEventRecord(h->ev[2], h->stream);
  if ((rc = run_pdf(h))) return rc;
  if (prof) hipEventRecord(h->ev[3], h->stream);
  if ((rc = run_dp_min(h))) return rc;
  if (prof) hipEventRecord(h->ev[4], h->stream);
  if ((rc = run_argmin_enqueue(h))) return rc;
  if (prof) hipEventRecord(h->ev[5], h->stream);
  return PBD_OK;
}

// A frame is ~40 launches whose arguments depend only on the frame geometry (the work tables are built once per
// geometry): with pbd_options.graph the second frame of a geometry is captured into a hipGraph (the first one
// runs eagerly: it also does the one-time per-device kernel attribute set-up, which is not a stream operation)
// and every later frame is ONE hipGraphLaunch.  The graph reads the frame from the handle's own image buffer, so
// an image that lives elsewhere in HBM is copied there first (0.9 MB, on the same stream).  Profiling runs (stage
// events) and level groups on extra streams use the eager path.
static int enqueue_all(pbd_handle* h, const uint8_t* d_src, int stride) {
  const bool graphable = h->opt.graph && !h->profiling;   // (frames of any depth: the launches depend on the plan only; round 5 replayed 8-bit plans only)
  if (!graphable || h->frames_on_plan == 0) {
    h->frames_on_plan++;
    return enqueue_stages(h, d_src, stride);
  }
  const size_t row = (size_t)h->fw * h->fcn * h->fesz;
  if (d_src != h->d_img) {
    if ((size_t)stride == row) HIPCHK(h, hipMemcpyAsync(h->d_img, d_src, row * h->fh * h->batch, hipMemcpyDeviceToDevice, h->stream));
    else HIPCHK(h, hipMemcpy2DAsync(h->d_img, row, d_src, stride, row, h->fh, hipMemcpyDeviceToDevice, h->stream));   // (strided sources: single frames only)
  }
  if (!h->gexec) {
    hipGraph_t graph = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_stages(h, h->d_img, (int)row);
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(h, PBD_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) { h->gexec = nullptr; return fail(h, PBD_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
  }
  HIPCHK(h, hipGraphLaunch(h->gexec, h->stream));
  h->frames_on_plan++;
  h->have_pyr = h->have_feat = h->have_resp = !h->compact;
  compact_mark_feat(h, false);
  compact_mark_resp(h, false);
  // k_hog's epilogue has rewritten the features' split copy; a compact plan's copy shares the x pass's pointer planes and is dead again
  h->feat_split_ok = h->split_parts != 0 && !h->compact;
  h->have_dp = true;
  h->min_ran = true;
  h->ext_ptr = false;
  h->root_dirty = false;
  h->dp_timed = false;
  h->pending = true;
  return PBD_OK;
}

static void read_stage_times(pbd_handle* h) {
  if (!h->profiling) return;
  for (int i = 0; i < 5; ++i) hipEventElapsedTime(&h->stage_ms[i], h->ev[i], h->ev[i + 1]);
  hipEventElapsedTime(&h->stage_ms[5], h->ev[0], h->ev[5]);
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
#pragma GCC visibility push(default)
extern "C" {

int pbd_create(const pbd_model_desc* model, const pbd_options* opt, pbd_handle** out) {
  if (!out) return PBD_ERR_ARG;
  *out = nullptr;
  pbd_handle* h = new (std::nothrow) pbd_handle();
  if (!h) return PBD_ERR_ARG;
  *out = h;  // returned even on failure so that pbd_last_error() can be read; destroy it either way
  pbd_options o{};
  if (opt) o = *opt;
  if (o.max_candidates <= 0) o.max_candidates = 4096;
  h->opt = o;
  if (o.scalar_type != PBD_SCALAR_F32 && o.scalar_type != PBD_SCALAR_F64) return fail(h, PBD_ERR_ARG, "scalar_type: PBD_SCALAR_F32 or PBD_SCALAR_F64");
  h->ts = (o.scalar_type == PBD_SCALAR_F64) ? 8 : 4;
  int rc = ingest_model(h, model);
  if (rc) return rc;
  if (o.reserved[0] < 0 || o.reserved[0] > 1024) return fail(h, PBD_ERR_ARG, "reserved[0] (nms_sz): 0 = off, or the window of the score-map NMS");
  h->nms_sz = o.reserved[0];
  h->conv_mode = o.conv_mode;
  if (h->conv_mode == PBD_CONV_AUTO)
    // measured on MI355X for N = 26 .. 312 5x5x32 filters at 640x480 (profiles/history/archive/r03b_conv_modes.json): the fp32 MFMA
    // implicit GEMM beats the direct VALU correlation at every N (26 filters: 0.11 vs 0.38 ms; 156: 0.40 vs 1.62;
    // 312: 0.76 vs 2.89) — the contraction is K = 800 deep whatever N is, so one 16-filter n-tile already pays.
    // The VALU kernel remains the bit-exact parity path (PBD_CONV_EXACT) and what banks of fewer than 16 filters get.
    // Any filter size goes the same way (3x3 .. 9x9: the contraction is kh * kw * 32 >= 288 deep; run-time tap loop of the same kernel).
    // Round 5: float handles take the split-product bank (k_conv_split.hip: the fp32 products as six exact bfloat16 partial
    // products on the bf16 matrix units, fp32 accumulators — errors of the fp32 MFMA chain's size, 2-3x its speed, and off the
    // vector ALU's pipe); a weight outside bfloat16's finite range (|w| >= 3e38: no trained model) keeps the fp32 MFMA bank.
    {
      bool splittable = h->ts == 4 && model->flen == PBD_FLEN;
      for (size_t i = 0; splittable && i < h->filters.size(); ++i) splittable = std::fabs(h->filters[i]) < 3.0e38f;
      h->conv_mode = model->nfilters >= 16 ? (splittable ? PBD_CONV_SPLIT : PBD_CONV_MFMA) : PBD_CONV_EXACT;
    }
  if ((h->conv_mode == PBD_CONV_SPLIT || h->conv_mode == PBD_CONV_SPLIT_F16) && (h->ts != 4 || model->flen != PBD_FLEN))
    return fail(h, PBD_ERR_UNSUPPORTED, "PBD_CONV_SPLIT / PBD_CONV_SPLIT_F16: float handles (32-channel HOG features)");
  if (h->conv_mode == PBD_CONV_SPLIT || h->conv_mode == PBD_CONV_SPLIT_F16)
    for (size_t i = 0; i < h->filters.size(); ++i)
      if (!(std::fabs(h->filters[i]) < 3.0e38f)) return fail(h, PBD_ERR_UNSUPPORTED, "PBD_CONV_SPLIT: a filter weight outside bfloat16's finite range");
  if (h->conv_mode < PBD_CONV_AUTO || h->conv_mode > PBD_CONV_SPLIT_F16) return fail(h, PBD_ERR_ARG, "conv_mode: PBD_CONV_*");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(h, PBD_ERR_HIP, "no HIP device visible");
  if (o.device < 0 || o.device >= ndev) return fail(h, PBD_ERR_ARG, "bad device ordinal");
  HIPCHK(h, hipSetDevice(o.device));
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  h->own_stream = true;
  for (int i = 0; i < 8; ++i) HIPCHK(h, hipEventCreate(&h->ev[i]));
  HIPCHK(h, hipEventCreate(&h->ev_dp0));
  HIPCHK(h, hipEventCreate(&h->ev_dp1));
  return upload_model(h);
}

int pbd_destroy(pbd_handle* h) {
  if (!h) return PBD_ERR_ARG;
  if (h->stream) hipStreamSynchronize(h->stream);
  free_frame(h);
  hipFree(h->d_wT); hipFree(h->d_wS); hipFree(h->d_split_oscale); hipFree(h->d_biasw); hipFree(h->d_hog_lut); hipFree(h->d_parent); hipFree(h->d_plane0); hipFree(h->d_nparts);
  hipFree(h->d_flat); hipFree(h->d_depth);
  hipFree(h->d_cand_count); hipFree(h->d_cand_rec); hipFree(h->d_cand_out);
  if (h->h_cand_out) hipHostFree(h->h_cand_out);
  if (h->h_cand_count) hipHostFree(h->h_cand_count);
  for (int i = 0; i < 8; ++i) if (h->ev[i]) hipEventDestroy(h->ev[i]);
  if (h->ev_dp0) hipEventDestroy(h->ev_dp0);
  if (h->ev_dp1) hipEventDestroy(h->ev_dp1);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
  return PBD_OK;
}

const char* pbd_last_error(const pbd_handle* h) { return h ? h->err.c_str() : "null handle"; }
int pbd_set_levels(pbd_handle* h, const int32_t* levels, int n) {
  if (!h || n < 0 || (n > 0 && !levels)) return PBD_ERR_ARG;
  std::vector<char> set;
  for (int i = 0; i < n; ++i) {
    if (levels[i] < 0 || levels[i] >= PBD_MAX_LEVELS) return fail(h, PBD_ERR_ARG, "level index out of range");
    if ((int)set.size() <= levels[i]) set.resize(levels[i] + 1, 0);
    set[levels[i]] = 1;
  }
  if (n > 0 && set.empty()) return fail(h, PBD_ERR_ARG, "empty level set");
  if (h->pending) return fail(h, PBD_ERR_STATE, "a frame is in flight: collect it first");
  hipStreamSynchronize(h->stream);
  h->level_set.swap(set);
  free_frame(h);   // the work tables are per geometry AND level set: re-planned on the next frame
  return PBD_OK;
}
int pbd_max_parts(const pbd_handle* h) { return h ? h->max_parts : 0; }

int pbd_set_stream(pbd_handle* h, void* s) {
  if (!h) return PBD_ERR_ARG;
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  h->stream = (hipStream_t)s;
  h->own_stream = false;
  return PBD_OK;
}

int pbd_detect_enqueue_dev_u8(pbd_handle* h, const void* d_im, int w, int hgt, int cn, int stride) {
  if (!h || !d_im) return PBD_ERR_ARG;
  if (h->pending) return fail(h, PBD_ERR_STATE, "previous frame not collected");
  if (stride < w * cn) return fail(h, PBD_ERR_ARG, "stride < w*cn");
  ON_DEVICE(h);
  int rc = plan_frame(h, w, hgt, cn);
  if (rc) return rc;
  return enqueue_all(h, (const uint8_t*)d_im, stride);
}

int pbd_detect_collect(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* count) {
  if (!h) return PBD_ERR_ARG;
  ON_DEVICE(h);
  int rc = collect(h, heads, boxes, locs, capacity, count);
  read_stage_times(h);
  return rc;
}

int pbd_detect_dev_u8(pbd_handle* h, const void* d_im, int w, int hgt, int cn, int stride, pbd_candidate_head* heads,
                      int32_t* boxes, int32_t* locs, int capacity, int* count) {
  int rc = pbd_detect_enqueue_dev_u8(h, d_im, w, hgt, cn, stride);
  if (rc) return rc;
  return pbd_detect_collect(h, heads, boxes, locs, capacity, count);
}

static int upload_image(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride);
extern "C++" int pbd_i_upload_image(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride) { return upload_image(h, im, w, hgt, cn, stride); }
static int upload_image(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride) {
  if (stride < w * cn) return fail(h, PBD_ERR_ARG, "stride < w*cn");
  ON_DEVICE(h);
  int rc = plan_frame(h, w, hgt, cn);
  if (rc) return rc;
  // tightly packed rows: one linear copy (a DMA-engine transfer when `im` is pinned); strided rows: a 2-D copy
  if (stride == w * cn) HIPCHK(h, hipMemcpyAsync(h->d_img, im, (size_t)w * cn * hgt, hipMemcpyHostToDevice, h->stream));
  else HIPCHK(h, hipMemcpy2DAsync(h->d_img, (size_t)w * cn, im, stride, (size_t)w * cn, hgt, hipMemcpyHostToDevice, h->stream));
  return PBD_OK;
}

int pbd_detect_enqueue_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride) {
  if (!h || !im) return PBD_ERR_ARG;
  if (h->pending) return fail(h, PBD_ERR_STATE, "previous frame not collected");
  int rc = upload_image(h, im, w, hgt, cn, stride);   // hipMemcpy2DAsync on the handle's stream, in front of the kernels
  if (rc) return rc;
  return enqueue_all(h, h->d_img, w * cn);
}

int pbd_detect_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride, pbd_candidate_head* heads,
                  int32_t* boxes, int32_t* locs, int capacity, int* count) {
  int rc = pbd_detect_enqueue_u8(h, im, w, hgt, cn, stride);
  if (rc) return rc;
  return pbd_detect_collect(h, heads, boxes, locs, capacity, count);
}

// ---- images of the other depths the reference accepts (src/HOGFeatures.cpp:136-146): single frames, host images, eager launches ----
static int upload_image_any(pbd_handle* h, const void* im, int depth, int w, int hgt, int cn, int stride) {
  const int esz = depth_esz(depth);
  if (!esz) return fail(h, PBD_ERR_UNSUPPORTED, "Unsupported image type (src/HOGFeatures.cpp:136-146: CV_8U, CV_16U, CV_32F, CV_64F)");
  const size_t row = (size_t)w * cn * esz;
  if (stride < 0 || (size_t)stride < row || stride % esz) return fail(h, PBD_ERR_ARG, "stride: bytes, >= w * cn * element size and a multiple of the element size");
  ON_DEVICE(h);
  int rc = plan_frame(h, w, hgt, cn, 1, depth);
  if (rc) return rc;
  if ((size_t)stride == row) HIPCHK(h, hipMemcpyAsync(h->d_img, im, row * hgt, hipMemcpyHostToDevice, h->stream));
  else HIPCHK(h, hipMemcpy2DAsync(h->d_img, row, im, stride, row, hgt, hipMemcpyHostToDevice, h->stream));
  return PBD_OK;
}
int pbd_detect_image(pbd_handle* h, const void* im, int depth, int w, int hgt, int cn, int stride, pbd_candidate_head* heads,
                     int32_t* boxes, int32_t* locs, int capacity, int* count) {
  if (!h || !im) return PBD_ERR_ARG;
  if (depth == PBD_DEPTH_8U) return pbd_detect_u8(h, (const uint8_t*)im, w, hgt, cn, stride, heads, boxes, locs, capacity, count);
  if (h->pending) return fail(h, PBD_ERR_STATE, "previous frame not collected");
  int rc = upload_image_any(h, im, depth, w, hgt, cn, stride);
  if (rc) return rc;
  if ((rc = enqueue_all(h, h->d_img, w * cn * h->fesz))) return rc;
  return pbd_detect_collect(h, heads, boxes, locs, capacity, count);
}
int pbd_pyramid_image(pbd_handle* h, const void* im, int depth, int w, int hgt, int cn, int stride) {
  if (!h || !im) return PBD_ERR_ARG;
  if (depth == PBD_DEPTH_8U) return pbd_pyramid_u8(h, (const uint8_t*)im, w, hgt, cn, stride);
  int rc = upload_image_any(h, im, depth, w, hgt, cn, stride);
  if (rc) return rc;
  if ((rc = run_image_pyramid(h, h->d_img, w * cn * h->fesz))) return rc;
  if ((rc = run_hog(h))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PBD_OK;
}

// ---- a batch of same-sized frames through ONE handle ---------------------------------------------------------
// (SURVEY 8b lists pbd_detect_batch_u8; configs[2] hands every GPU 4 frames.)  The frames of a batch are planned as
// extra "virtual levels" (pbd_internal.hpp), so every stage is one launch — or one chain of launches — for the whole
// batch: the same kernels, B times the blocks per launch.  Results per frame are identical to pbd_detect_u8.
int pbd_detect_batch_enqueue_dev_u8(pbd_handle* h, const void* d_ims, int nframes, int w, int hgt, int cn) {
  if (!h || !d_ims || nframes < 1) return PBD_ERR_ARG;
  if (h->pending) return fail(h, PBD_ERR_STATE, "previous frame not collected");
  ON_DEVICE(h);
  int rc = plan_frame(h, w, hgt, cn, nframes);
  if (rc) return rc;
  return enqueue_all(h, (const uint8_t*)d_ims, w * cn);
}
int pbd_detect_batch_enqueue_u8(pbd_handle* h, const uint8_t* const* ims, int nframes, int w, int hgt, int cn, int stride) {
  if (!h || !ims || nframes < 1) return PBD_ERR_ARG;
  if (h->pending) return fail(h, PBD_ERR_STATE, "previous frame not collected");
  if (stride < w * cn) return fail(h, PBD_ERR_ARG, "stride < w*cn");
  ON_DEVICE(h);
  int rc = plan_frame(h, w, hgt, cn, nframes);
  if (rc) return rc;
  const size_t fb = (size_t)w * cn * hgt;
  for (int f = 0; f < nframes; ++f) {
    if (!ims[f]) return fail(h, PBD_ERR_ARG, "null frame pointer");
    if (stride == w * cn) HIPCHK(h, hipMemcpyAsync(h->d_img + fb * f, ims[f], fb, hipMemcpyHostToDevice, h->stream));
    else HIPCHK(h, hipMemcpy2DAsync(h->d_img + fb * f, (size_t)w * cn, ims[f], stride, (size_t)w * cn, hgt, hipMemcpyHostToDevice, h->stream));
  }
  return enqueue_all(h, h->d_img, w * cn);
}
// frame f's candidates at heads[f * capacity], boxes[f * capacity * max_parts * 4], locs[f * capacity * max_parts * 3];
// counts[f] = number found in frame f (PBD_ERR_CAPACITY if any exceeds `capacity`)
int pbd_detect_batch_collect(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* counts) {
  if (!h || !heads || !counts || capacity < 0) return PBD_ERR_ARG;
  if (!h->pending) return fail(h, PBD_ERR_STATE, "collect without a pending detect");
  if (h->d_gsend) return fail(h, PBD_ERR_STATE, "handle belongs to an RCCL-gathering pbd_group: collect through the group");
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const int found = h->h_cand_count[0];
  int rc = pbd_i_finish_frame(h, found);
  read_stage_times(h);
  if (rc) return rc;
  const int mp = h->max_parts, n1 = h->nlevels, B = h->batch;
  std::vector<std::vector<const char*>> per(B);
  for (int i = 0; i < found; ++i) {
    const char* r = h->h_cand_out + h->cand_stride * i;
    per[((const pbd_candidate_head*)r)->level / n1].push_back(r);
  }
  int status = PBD_OK;
  for (int f = 0; f < B; ++f) {
    counts[f] = (int)per[f].size();
    pbd_candidate_head* hf = heads + (size_t)f * capacity;
    rc = pbd_i_emit(h, per[f], hf, boxes ? boxes + (size_t)f * capacity * mp * 4 : nullptr, locs ? locs + (size_t)f * capacity * mp * 3 : nullptr, capacity);
    if (rc == PBD_ERR_CAPACITY) { status = rc; continue; }
    if (rc) return rc;
    for (int i = 0; i < counts[f]; ++i) hf[i].level -= f * n1;   // virtual level -> the frame's own pyramid level
  }
  return status;
}
int pbd_detect_batch_u8(pbd_handle* h, const uint8_t* const* ims, int nframes, int w, int hgt, int cn, int stride,
                        pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* counts) {
  int rc = pbd_detect_batch_enqueue_u8(h, ims, nframes, w, hgt, cn, stride);
  if (rc) return rc;
  return pbd_detect_batch_collect(h, heads, boxes, locs, capacity, counts);
}

// ---- the planner's one measured rule, re-measured on the caller's own frames ---------------------------------
// The distance transform's block geometry (lanes and LDS per block) changes the stage's time by a few per cent either way depending on
// the frame size, the model and whether frames come singly or in batches, and never its results (dt_core.hpp: the same exact
// algorithm under any segmentation).  plan_frame picks by a rule measured on seven sizes with the person model;
// pbd_tune_plan runs the caller's frame through both geometries of a float handle — `batch` copies per call, 2 warm-up + 3 timed
// calls each, the dp_min stage's GPU time from the stage events — and keeps the faster one for every later plan of this handle.
// im == NULL: back to the rule.  chosen: 1 = 256 lanes / 40 KB, 2 = 128 lanes / 25 KB, 0 = nothing to choose (double handles).
int pbd_tune_plan(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride, int batch, int* chosen, double ms[2]) {
  if (!h) return PBD_ERR_ARG;
  if (h->pending) return fail(h, PBD_ERR_STATE, "previous frame not collected");
  // (a member of an RCCL-gathering group collects through the group only: the measurement's own collect() would fail with the frame
  //  left pending — ADVICE r05)
  if (h->d_gsend) return fail(h, PBD_ERR_STATE, "handle belongs to an RCCL-gathering pbd_group: tune a handle of its own");
  if (chosen) *chosen = 0;
  if (ms) ms[0] = ms[1] = 0.0;
  ON_DEVICE(h);
  // the plan is dropped as a whole (buffers, stage flags, captured graph): a stage getter after pbd_tune_plan answers "no frame
  // geometry" AND the stage flags say so — nothing can run on the last measured geometry
  auto drop_plan = [&]() { if (h->stream) hipStreamSynchronize(h->stream); h->pending = false; free_frame(h); };
  if (!im) { h->dt_geom = 0; drop_plan(); return PBD_OK; }
  if (h->ts != 4) return PBD_OK;
  if (batch < 1 || batch > 64) return fail(h, PBD_ERR_ARG, "batch: 1..64 frames");
  const int cap = h->opt.max_candidates;            // per frame (pbd_detect_batch_collect: heads[batch][capacity])
  std::vector<pbd_candidate_head> heads((size_t)cap * batch);
  std::vector<int> counts((size_t)batch);
  std::vector<const uint8_t*> ims((size_t)batch, im);
  const bool prof = h->profiling;
  const int before = h->dt_geom;
  float stage_before[6];
  for (int i = 0; i < 6; ++i) stage_before[i] = h->stage_ms[i];
  double t[2] = {0, 0};
  int rc = PBD_OK;
  h->profiling = true;
  for (int g = 1; g <= 2 && !rc; ++g) {
    h->dt_geom = g; drop_plan();                    // (plan_frame re-plans under the geometry)
    double v[3] = {0, 0, 0};
    for (int i = 0; i < 5 && !rc; ++i) {
      rc = batch == 1 ? pbd_detect_u8(h, im, w, hgt, cn, stride, heads.data(), nullptr, nullptr, cap, counts.data())
                      : pbd_detect_batch_u8(h, ims.data(), batch, w, hgt, cn, stride, heads.data(), nullptr, nullptr, cap, counts.data());
      if (rc == PBD_ERR_CAPACITY) rc = PBD_OK;      // (a low threshold: the stage times are what is wanted)
      if (i >= 2) v[i - 2] = h->stage_ms[3];
    }
    std::sort(v, v + 3);
    t[g - 1] = v[1];
  }
  h->profiling = prof;
  for (int i = 0; i < 6; ++i) h->stage_ms[i] = stage_before[i];   // the caller's last stage times, not the measurement's
  const std::string err = h->err;
  h->dt_geom = rc ? before : (t[0] <= t[1] ? 1 : 2);
  drop_plan();                                      // whatever happened: nothing in flight, no plan of the measurement's call shape left behind
  if (rc) { h->err = err; return rc; }
  if (chosen) *chosen = h->dt_geom;
  if (ms) { ms[0] = t[0]; ms[1] = t[1]; }
  return PBD_OK;
}

// ---- stage entry points -----------------------------------------------------
int pbd_pyramid_geometry(const pbd_handle* h, int w, int hgt, int* nlevels, int32_t* img_w, int32_t* img_h,
                         int32_t* cell_w, int32_t* cell_h, float* scales) {
  if (!h || !nlevels) return PBD_ERR_ARG;
  static thread_local Level lv[PBD_MAX_LEVELS];
  int n = 0;
  if (w < 3 || hgt < 3 || compute_geometry(w, hgt, h->md.sbin, h->md.interval, &n, lv)) return PBD_ERR_ARG;
  *nlevels = n;
  for (int l = 0; l < n; ++l) {
    if (img_w) img_w[l] = lv[l].iw;
    if (img_h) img_h[l] = lv[l].ih;
    if (cell_w) cell_w[l] = lv[l].cw;
    if (cell_h) cell_h[l] = lv[l].ch;
    if (scales) scales[l] = lv[l].scale;
  }
  return PBD_OK;
}

int pbd_begin_frame(pbd_handle* h, int w, int hgt, int cn) {
  if (!h) return PBD_ERR_ARG;
  ON_DEVICE(h);
  int rc = plan_frame(h, w, hgt, cn);
  if (rc) return rc;
  h->have_pyr = h->have_feat = h->have_resp = h->have_dp = false;
  h->min_ran = false;
  compact_mark_feat(h, false);
  compact_mark_resp(h, false);
  h->ext_set.clear(); h->root_set.clear();
  return PBD_OK;
}

int pbd_pyramid_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride) {
  if (!h || !im) return PBD_ERR_ARG;
  int rc = upload_image(h, im, w, hgt, cn, stride);
  if (rc) return rc;
  if ((rc = run_image_pyramid(h, h->d_img, w * cn))) return rc;
  if ((rc = run_hog(h))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PBD_OK;
}

#define CHECK_LEVEL(h, level)                                                     \
  if (!(h)) return PBD_ERR_ARG;                                                   \
  if ((h)->fw == 0) return fail(h, PBD_ERR_STATE, "no frame geometry");           \
  if ((h)->batch > 1) return fail(h, PBD_ERR_STATE, "the current plan is a batch of frames (pbd_detect_batch_*): the stage entry points address single-frame plans"); \
  if ((level) < 0 || (level) >= (h)->nlevels) return fail(h, PBD_ERR_ARG, "level out of range");

int pbd_get_level_image(pbd_handle* h, int level, uint8_t* out) {
  CHECK_LEVEL(h, level);
  if (!h->have_pyr) return fail(h, PBD_ERR_STATE, "pyramid not computed");
  const Level& L = h->lv[level];
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->fdepth != PBD_DEPTH_8U) return fail(h, PBD_ERR_STATE, "the planned frame is not 8-bit: pbd_get_level_image_raw");
  HIPCHK(h, hipMemcpy(out, h->d_pyr + L.img_off, (size_t)L.iw * L.ih * h->fcn, hipMemcpyDeviceToHost));
  return PBD_OK;
}
int pbd_get_level_image_raw(pbd_handle* h, int level, void* out, size_t out_bytes) {
  CHECK_LEVEL(h, level);
  if (!out) return PBD_ERR_ARG;
  if (!h->have_pyr) return fail(h, PBD_ERR_STATE, "pyramid not computed");
  const Level& L = h->lv[level];
  const size_t bytes = (size_t)L.iw * L.ih * h->fcn * h->fesz;
  if (out_bytes < bytes) return fail(h, PBD_ERR_CAPACITY, "pbd_get_level_image_raw: iw * ih * cn * element size bytes");
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_pyr + L.img_off, bytes, hipMemcpyDeviceToHost));
  return PBD_OK;
}
// The float / _f64 entry points share one body; a handle only answers the pair that matches the
// instantiation it was created for (cv::Mat depth CV_32F vs CV_64F in the reference).
#define CHECK_SCALAR(h, want) \
  do { if ((h)->ts != (want)) return fail((h), PBD_ERR_STATE, (want) == 4 ? "handle is PartsBasedDetector<double>: use the _f64 entry point" \
                                                                         : "handle is PartsBasedDetector<float>: use the float entry point"); } while (0)
static int get_level_features_(pbd_handle* h, int level, void* out, int ts) {
  CHECK_LEVEL(h, level);
  CHECK_SCALAR(h, ts);
  if (!h->have_feat) return fail(h, PBD_ERR_STATE, "features not computed");
  const Level& L = h->lv[level];
  if (!L.active) return fail(h, PBD_ERR_STATE, "level is not processed by this handle (pbd_set_levels / level_begin..level_end)");
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_feat + L.cell_off * PBD_FLEN * ts, (size_t)L.cw * L.ch * PBD_FLEN * ts, hipMemcpyDeviceToHost));
  return PBD_OK;
}
static int set_level_features_(pbd_handle* h, int level, const void* in, int ts) {
  CHECK_LEVEL(h, level);
  CHECK_SCALAR(h, ts);
  const Level& L = h->lv[level];
  if (!in) return fail(h, PBD_ERR_ARG, "null feature matrix");
  // The split-product banks carry a feature as bfloat16 / binary16 parts: a value outside the parts' finite range would become inf / NaN parts
  // and poison every response the MFMA sums it into, where the EXACT / MFMA banks propagate it as ordinary fp32.  HOG features
  // are in [0, 0.4]; a caller's own features are held to the bank's domain here, on the host copy that is being uploaded anyway.
  if (ts == 4 && (h->conv_mode == PBD_CONV_SPLIT || h->conv_mode == PBD_CONV_SPLIT_F16)) {
    const float lim = h->conv_mode == PBD_CONV_SPLIT_F16 ? 16.0f : 3.0e38f;
    const float* f = (const float*)in;
    const size_t nf = (size_t)L.cw * L.ch * PBD_FLEN;
    for (size_t i = 0; i < nf; ++i)
      if (!(std::fabs(f[i]) < lim))
        return fail(h, PBD_ERR_ARG, h->conv_mode == PBD_CONV_SPLIT_F16 ? "PBD_CONV_SPLIT_F16: a feature outside the bank's domain (|f| < 16, finite)"
                                                                        : "PBD_CONV_SPLIT: a feature outside bfloat16's finite range (|f| < 3e38, finite): use PBD_CONV_MFMA / PBD_CONV_EXACT");
  }
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(h->d_feat + L.cell_off * PBD_FLEN * ts, in, (size_t)L.cw * L.ch * PBD_FLEN * ts, hipMemcpyHostToDevice));
  h->feat_split_ok = false;   // (the split-product bank's copy of the features is re-derived in front of the next pdf())
  if (h->compact) {   // the write went over the Ik planes / the x pass's scratch; the other levels may still be stale
    h->have_dp = h->min_ran = false;
    if (h->feat_ok.size() != (size_t)h->nvl) compact_mark_feat(h, false);
    h->feat_ok[level] = 1;
    h->have_feat = all_active_set(h, h->feat_ok, 1);
  } else {
    h->have_feat = true;
  }
  return PBD_OK;
}
int pbd_get_level_features(pbd_handle* h, int level, float* out) { return get_level_features_(h, level, out, 4); }
int pbd_get_level_features_f64(pbd_handle* h, int level, double* out) { return get_level_features_(h, level, out, 8); }
int pbd_set_level_features(pbd_handle* h, int level, const float* in) { return set_level_features_(h, level, in, 4); }
int pbd_set_level_features_f64(pbd_handle* h, int level, const double* in) { return set_level_features_(h, level, in, 8); }
int pbd_pdf(pbd_handle* h) {
  if (!h) return PBD_ERR_ARG;
  if (!h->have_feat) return fail(h, PBD_ERR_STATE, h->compact ? "pdf(): features are not resident (compact memory plan: min() reuses their memory — run pyramid() or hand in every level again)" : "pdf() before pyramid()");
  ON_DEVICE(h);
  int rc = run_pdf(h);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PBD_OK;
}
static int get_level_response_(pbd_handle* h, int level, int filter, void* out, int ts) {
  CHECK_LEVEL(h, level);
  CHECK_SCALAR(h, ts);
  if (!h->have_resp) return fail(h, PBD_ERR_STATE, "responses not computed");
  if (filter < 0 || filter >= h->md.nfilters) return fail(h, PBD_ERR_ARG, "filter out of range");
  const Level& L = h->lv[level];
  if (!L.active) return fail(h, PBD_ERR_STATE, "level is not processed by this handle (pbd_set_levels / level_begin..level_end)");
  const size_t HW = (size_t)L.cw * L.ch;
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_resp + (L.cell_off * h->md.nfilters + filter * HW) * ts, HW * ts, hipMemcpyDeviceToHost));
  return PBD_OK;
}
static int set_level_response_(pbd_handle* h, int level, int filter, const void* in, int ts) {
  CHECK_LEVEL(h, level);
  CHECK_SCALAR(h, ts);
  if (filter < 0 || filter >= h->md.nfilters) return fail(h, PBD_ERR_ARG, "filter out of range");
  const Level& L = h->lv[level];
  const size_t HW = (size_t)L.cw * L.ch;
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(h->d_resp + (L.cell_off * h->md.nfilters + filter * HW) * ts, in, HW * ts, hipMemcpyHostToDevice));
  if (h->compact) {   // min() transformed the planes in place: the responses are valid again once EVERY plane has been handed in (or pdf() re-run)
    if (h->resp_ok.size() != (size_t)h->nvl * h->md.nfilters) compact_mark_resp(h, false);
    h->resp_ok[(size_t)level * h->md.nfilters + filter] = 1;
    h->have_resp = all_active_set(h, h->resp_ok, h->md.nfilters);
  } else {
    h->have_resp = true;
  }
  return PBD_OK;
}
int pbd_get_level_response(pbd_handle* h, int level, int filter, float* out) { return get_level_response_(h, level, filter, out, 4); }
int pbd_get_level_response_f64(pbd_handle* h, int level, int filter, double* out) { return get_level_response_(h, level, filter, out, 8); }
int pbd_set_level_response(pbd_handle* h, int level, int filter, const float* in) { return set_level_response_(h, level, filter, in, 4); }
int pbd_set_level_response_f64(pbd_handle* h, int level, int filter, const double* in) { return set_level_response_(h, level, filter, in, 8); }
int pbd_dp_min(pbd_handle* h) {
  if (!h) return PBD_ERR_ARG;
  if (!h->have_resp) return fail(h, PBD_ERR_STATE, h->compact ? "min(): responses are not resident (compact memory plan: min() transforms them in place — run pdf() or hand in every plane again)" : "min() before pdf()");
  ON_DEVICE(h);
  int rc = run_dp_min(h);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PBD_OK;
}
int pbd_get_dp_pointers(pbd_handle* h, int level, int component, int part, int parent_mix, int32_t* ix, int32_t* iy, int32_t* ik) {
  CHECK_LEVEL(h, level);
  if (!h->have_dp) return fail(h, PBD_ERR_STATE, "min() not run");
  if (component < 0 || component >= h->md.ncomponents) return fail(h, PBD_ERR_ARG, "component out of range");
  const int p0 = h->part_offset[component], cnp = h->part_offset[component + 1] - p0;
  if (part < 1 || part >= cnp) return fail(h, PBD_ERR_ARG, "part out of range (1..nparts-1)");
  const PartInfo& P = h->parts[p0 + part];
  const int L_ = h->parts[p0 + P.parent].K;
  if (parent_mix < 0 || parent_mix >= L_) return fail(h, PBD_ERR_ARG, "parent mixture out of range");
  const Level& L = h->lv[level];
  const size_t HW = (size_t)L.cw * L.ch;
  const size_t po = L.cell_off * h->nplanes + (size_t)(P.plane0 + parent_mix) * HW;
  if (!L.active) return fail(h, PBD_ERR_STATE, "level is not processed by this handle");
  // Ik is stored; Ix / Iy (reducePickIndex of the composed DT pointers, src/DynamicProgram.cpp:144-149) are
  // composed here from the DT pointer planes of the winning mixture, exactly as k_backtrack does per candidate
  std::vector<uint8_t> c(HW);
  std::vector<int16_t> X((size_t)P.K * HW), Y((size_t)P.K * HW);
  const size_t so = (size_t)h->scr_base[(size_t)level * h->parts.size() + (p0 + part)];
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(c.data(), h->d_pk + po, HW, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(X.data(), h->d_dt_ixT + so, X.size() * 2, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(Y.data(), h->d_dt_iy + so, Y.size() * 2, hipMemcpyDeviceToHost));
  const int W = L.cw;
  for (size_t i = 0; i < HW; ++i) {
    const size_t mo = (size_t)c[i] * HW;
    const int m_ = (int)(i / W), n_ = (int)(i - (size_t)m_ * W);
    int x, y;
    if (!h->opt.dt_correct_ptr) { x = X[mo + i]; y = Y[mo + (size_t)m_ * W + x]; }
    else { y = Y[mo + i]; x = X[mo + (size_t)y * W + n_]; }
    if (ix) ix[i] = x;
    if (iy) iy[i] = y;
    if (ik) ik[i] = c[i];
  }
  return PBD_OK;
}
static int get_root_(pbd_handle* h, int level, int component, void* rootv, int32_t* rooti, int ts) {
  CHECK_LEVEL(h, level);
  CHECK_SCALAR(h, ts);
  if (!h->have_dp) return fail(h, PBD_ERR_STATE, "min() not run");
  if (component < 0 || component >= h->md.ncomponents) return fail(h, PBD_ERR_ARG, "component out of range");
  const Level& L = h->lv[level];
  if (!L.active) return fail(h, PBD_ERR_STATE, "level is not processed by this handle (pbd_set_levels / level_begin..level_end)");
  const size_t HW = (size_t)L.cw * L.ch;
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (rootv) HIPCHK(h, hipMemcpy(rootv, h->d_rootv + (L.cell_off * h->md.ncomponents + component * HW) * ts, HW * ts, hipMemcpyDeviceToHost));
  if (rooti) HIPCHK(h, hipMemcpy(rooti, h->d_rooti + L.cell_off * h->md.ncomponents + component * HW, HW * 4, hipMemcpyDeviceToHost));
  return PBD_OK;
}
int pbd_get_root(pbd_handle* h, int level, int component, float* rootv, int32_t* rooti) { return get_root_(h, level, component, rootv, rooti, 4); }
int pbd_get_root_f64(pbd_handle* h, int level, int component, double* rootv, int32_t* rooti) { return get_root_(h, level, component, rootv, rooti, 8); }
// DynamicProgram<T>::argmin takes rootv / rooti / Ix / Iy / Ik as ARGUMENTS (include/DynamicProgram.hpp:75): a caller
// that hands it tables other than the ones this handle's min() left on the device injects them here.
static int set_root_(pbd_handle* h, int level, int component, const void* rootv, const int32_t* rooti, int ts) {
  CHECK_LEVEL(h, level);
  CHECK_SCALAR(h, ts);
  if (component < 0 || component >= h->md.ncomponents) return fail(h, PBD_ERR_ARG, "component out of range");
  const Level& L = h->lv[level];
  if (!L.active) return fail(h, PBD_ERR_STATE, "level is not processed by this handle (pbd_set_levels / level_begin..level_end)");
  const size_t HW = (size_t)L.cw * L.ch;
  if (rooti) {   // rooti becomes the root's mixture in back-tracking and indexes the children's Ik planes
    const int K0 = h->parts[h->part_offset[component]].K;
    for (size_t i = 0; i < HW; ++i)
      if (rooti[i] < 0 || rooti[i] >= K0) return fail(h, PBD_ERR_ARG, "rooti entry out of range (0..K_root-1)");
  }
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (rootv) HIPCHK(h, hipMemcpy(h->d_rootv + (L.cell_off * h->md.ncomponents + component * HW) * ts, rootv, HW * ts, hipMemcpyHostToDevice));
  if (rooti) HIPCHK(h, hipMemcpy(h->d_rooti + L.cell_off * h->md.ncomponents + component * HW, rooti, HW * 4, hipMemcpyHostToDevice));
  h->root_dirty = true;
  if (rootv && rooti) {
    if (h->root_set.size() != (size_t)h->nvl * h->md.ncomponents) h->root_set.assign((size_t)h->nvl * h->md.ncomponents, 0);
    h->root_set[(size_t)level * h->md.ncomponents + component] = 1;
  }
  return PBD_OK;
}
int pbd_set_root(pbd_handle* h, int level, int component, const float* rootv, const int32_t* rooti) { return set_root_(h, level, component, rootv, rooti, 4); }
int pbd_set_root_f64(pbd_handle* h, int level, int component, const double* rootv, const int32_t* rooti) { return set_root_(h, level, component, rootv, rooti, 8); }
int pbd_set_dp_pointers(pbd_handle* h, int level, int component, int part, int parent_mix, const int32_t* ix, const int32_t* iy, const int32_t* ik) {
  CHECK_LEVEL(h, level);
  if (!ix || !iy || !ik) return fail(h, PBD_ERR_ARG, "null table");
  if (component < 0 || component >= h->md.ncomponents) return fail(h, PBD_ERR_ARG, "component out of range");
  const int p0 = h->part_offset[component], cnp = h->part_offset[component + 1] - p0;
  if (part < 1 || part >= cnp) return fail(h, PBD_ERR_ARG, "part out of range (1..nparts-1)");
  const PartInfo& P = h->parts[p0 + part];
  if (parent_mix < 0 || parent_mix >= h->parts[p0 + P.parent].K) return fail(h, PBD_ERR_ARG, "parent mixture out of range");
  const Level& L = h->lv[level];
  if (!L.active) return fail(h, PBD_ERR_STATE, "level is not processed by this handle");
  const size_t HW = (size_t)L.cw * L.ch;
  ON_DEVICE(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!h->d_extx) {   // first use on this frame plan: composed planes for every (level, plane), 2 x int16
    const size_t n = h->cells * (size_t)std::max(h->nplanes, 1);
    int rc;
    if ((rc = dev_alloc(h, &h->d_extx, n))) return rc;
    if ((rc = dev_alloc(h, &h->d_exty, n))) return rc;
    HIPCHK(h, hipMemset(h->d_extx, 0, n * sizeof(int16_t)));   // planes never handed in read as (0, 0): in range for every level
    HIPCHK(h, hipMemset(h->d_exty, 0, n * sizeof(int16_t)));
    std::vector<unsigned long long> base((size_t)h->nvl * h->md.ncomponents);
    for (int l = 0; l < h->nvl; ++l)
      for (int c = 0; c < h->md.ncomponents; ++c)
        base[(size_t)l * h->md.ncomponents + c] = h->lv[l].cell_off * h->nplanes + (size_t)h->comp_plane0[c] * h->lv[l].cw * h->lv[l].ch;
    if ((rc = dev_upload(h, &h->d_ext_base, base))) return rc;
  }
  if (!h->ext_ptr && h->have_dp && h->min_ran) {
    // the planes the caller does NOT hand in keep this handle's own tables: materialise all of them once, composed
    // (costly, but this is the compatibility path of a mixed-engine caller, not detect())
    std::vector<int32_t> x, y, k;
    std::vector<int16_t> xs, ys;
    for (int l = 0; l < h->nlevels; ++l) {
      const Level& LL = h->lv[l];
      const size_t hw = (size_t)LL.cw * LL.ch;
      if (!LL.active || hw == 0) continue;
      x.resize(hw); y.resize(hw); k.resize(hw); xs.resize(hw); ys.resize(hw);
      for (int c = 0; c < h->md.ncomponents; ++c) {
        const int q0 = h->part_offset[c], qn = h->part_offset[c + 1] - q0;
        for (int pp = 1; pp < qn; ++pp)
          for (int m2 = 0; m2 < h->parts[q0 + h->parts[q0 + pp].parent].K; ++m2) {
            int rc = pbd_get_dp_pointers(h, l, c, pp, m2, x.data(), y.data(), k.data());
            if (rc) return rc;
            for (size_t i = 0; i < hw; ++i) { xs[i] = (int16_t)x[i]; ys[i] = (int16_t)y[i]; }
            const size_t eo = LL.cell_off * h->nplanes + (size_t)(h->parts[q0 + pp].plane0 + m2) * hw;
            HIPCHK(h, hipMemcpy(h->d_extx + eo, xs.data(), hw * 2, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(h->d_exty + eo, ys.data(), hw * 2, hipMemcpyHostToDevice));
          }
      }
    }
  }
  std::vector<int16_t> xs(HW), ys(HW);
  std::vector<uint8_t> ks(HW);
  for (size_t i = 0; i < HW; ++i) {
    if (ix[i] < 0 || ix[i] >= L.cw || iy[i] < 0 || iy[i] >= L.ch || ik[i] < 0 || ik[i] >= P.K)
      return fail(h, PBD_ERR_ARG, "pointer table entry out of range");
    xs[i] = (int16_t)ix[i]; ys[i] = (int16_t)iy[i]; ks[i] = (uint8_t)ik[i];
  }
  const size_t eo = L.cell_off * h->nplanes + (size_t)(P.plane0 + parent_mix) * HW;
  HIPCHK(h, hipMemcpy(h->d_extx + eo, xs.data(), HW * 2, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_exty + eo, ys.data(), HW * 2, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_pk + eo, ks.data(), HW, hipMemcpyHostToDevice));
  h->ext_ptr = true;
  if (h->compact) {   // d_pk is the memory of the level images + features (and the first call's d_extx / d_exty allocation moved nothing, but the
                      // Ik write above went over them): pyramid() -> set_dp_pointers -> pdf() must not run on clobbered features
    h->have_pyr = h->have_feat = false;
    compact_mark_feat(h, false);
  }
  if (h->ext_set.size() != (size_t)h->nvl * std::max(h->nplanes, 1)) h->ext_set.assign((size_t)h->nvl * std::max(h->nplanes, 1), 0);
  h->ext_set[(size_t)level * std::max(h->nplanes, 1) + P.plane0 + parent_mix] = 1;
  // with a min() of this handle behind them the planes not handed in keep its tables; without one, back-tracking may only
  // run once EVERY plane and every root table of the active levels has been provided (pbd_dp_argmin checks)
  h->have_dp = true;
  return PBD_OK;
}
int pbd_get_footprint(const pbd_handle* h, size_t* frame_bytes, size_t* model_bytes) {
  if (!h) return PBD_ERR_ARG;
  if (frame_bytes) *frame_bytes = h->frame_bytes;
  if (model_bytes) *model_bytes = h->model_bytes;
  return PBD_OK;
}
int pbd_abi_version(void) { return PBD_ABI_VERSION; }
int pbd_get_conv_mode(const pbd_handle* h) { return h ? h->conv_mode : -PBD_ERR_ARG; }   // what PBD_CONV_AUTO resolved to (negative: error)
int pbd_get_stage_state(const pbd_handle* h, int32_t state[4]) {
  if (!h || !state) return PBD_ERR_ARG;
  state[0] = h->have_pyr; state[1] = h->have_feat; state[2] = h->have_resp; state[3] = h->have_dp;
  return PBD_OK;
}

int pbd_dp_argmin(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* count) {
  if (!h) return PBD_ERR_ARG;
  if (!h->have_dp) return fail(h, PBD_ERR_STATE, "argmin() before min()");
  if (h->batch > 1) return fail(h, PBD_ERR_STATE, "the current plan is a batch of frames: its candidates come from pbd_detect_batch_collect");
  if (h->pending) return fail(h, PBD_ERR_STATE, "previous frame not collected");
  if (!h->min_ran) {   // tables handed in by the caller only: all of them, or back-tracking would walk uninitialised planes
    if (!all_active_set(h, h->ext_set, std::max(h->nplanes, 1)) && h->nplanes > 0)
      return fail(h, PBD_ERR_STATE, "argmin(): no min() on this frame and not every pointer table has been handed in (pbd_set_dp_pointers)");
    if (!all_active_set(h, h->root_set, h->md.ncomponents))
      return fail(h, PBD_ERR_STATE, "argmin(): no min() on this frame and not every root table has been handed in (pbd_set_root)");
  }
  ON_DEVICE(h);
  int rc = run_argmin_enqueue(h);
  if (rc) { h->pending = false; return rc; }
  return collect(h, heads, boxes, locs, capacity, count);
}

// ---- stand-alone primitives -------------------------------------------------
static int dt2d_(pbd_handle* h, const void* in, int rows, int cols, double ax, double bx, double ay, double by, int osx,
                 int osy, void* out, int32_t* ix, int32_t* iy, int tsz) {
  if (!h || !in || rows <= 0 || cols <= 0 || rows > 32767 || cols > 32767) return PBD_ERR_ARG;
  CHECK_SCALAR(h, tsz);
  if (ax == 0 || ay == 0) return fail(h, PBD_ERR_ARG, "a must be non-zero");
  ON_DEVICE(h);
  const size_t HW = (size_t)rows * cols;
  const size_t ts = (size_t)tsz;
  char *d_in, *d_tmp, *d_sdt;
  int16_t *d_ixT, *d_iy;
  HIPCHK(h, hipMalloc(&d_in, HW * ts)); HIPCHK(h, hipMalloc(&d_tmp, HW * ts)); HIPCHK(h, hipMalloc(&d_sdt, HW * ts));
  HIPCHK(h, hipMalloc(&d_ixT, HW * 2)); HIPCHK(h, hipMalloc(&d_iy, HW * 2));
  HIPCHK(h, hipMemcpyAsync(d_in, in, HW * ts, hipMemcpyHostToDevice, h->stream));
  DtMap maps[2] = {dt_map(d_in, d_tmp, d_ixT, 0.f, 0.f, osx, 1), dt_map(d_tmp, d_sdt, d_iy, 0.f, 0.f, osy, 0)};
  maps[0].a = ax; maps[0].b = bx; maps[0].r2a = 1.0 / (2.0 * ax);   // the caller's quadratics (not the model's -w)
  maps[1].a = ay; maps[1].b = by; maps[1].r2a = 1.0 / (2.0 * ay);
  // block size of THIS call's two launches: a local — the handle's dt_nt belongs to the frame plan, whose uploaded tasks carry
  // the geometry derived from it (nsub, P, chunk); overwriting it here made a detect() after a pbd_dt2d launch 128-lane blocks
  // over 256-lane tasks
  int nt = tsz == 8 ? 64 : PBD_DT_NT_DEFAULT;
  if (const char* e = PBD_PROBE_ENV("PBD_DT_NT")) nt = std::max(64, std::min(256, atoi(e) & ~63));
  size_t dt_base = 40 * 1024;
  if (const char* e = PBD_PROBE_ENV("PBD_DT_BUDGET_KB")) dt_base = (size_t)atoi(e) * 1024;
  const size_t budget = std::max<size_t>(dt_base, dt_lds_bytes(dt_stride_for(std::max(rows, cols)), 4, tsz, nt));
  if (budget > 160 * 1024) return fail(h, PBD_ERR_UNSUPPORTED, "map too large for the LDS-resident distance transform");
  DtGroup groups[2] = {dt_group(0, 1, rows, cols, budget, tsz, nt, h->dt_seg), dt_group(1, 1, cols, rows, budget, tsz, nt, h->dt_seg)};
  std::vector<DtTask> tasks;
  dt_add_tasks(groups[0], tasks);
  const int nx = (int)tasks.size();
  dt_add_tasks(groups[1], tasks);
  dt_mark_fused(tasks, maps, tsz);   // (the caller's quadratics: fused arithmetic only if they are converted floats)
  DtMap* d_maps; DtTask* d_tasks;
  HIPCHK(h, hipMalloc(&d_maps, sizeof(maps)));
  HIPCHK(h, hipMalloc(&d_tasks, sizeof(DtTask) * tasks.size()));
  HIPCHK(h, hipMemcpyAsync(d_maps, maps, sizeof(maps), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(d_tasks, tasks.data(), sizeof(DtTask) * tasks.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // host staging buffers above are pageable
  launch_dt_pass(d_tasks, nx, d_maps, nullptr, nullptr, h->d_biasw, budget, tsz, nt, 0, h->stream);
  if (PBD_PROBE_ENV("PBD_DEBUG_SKIP_Y")) { hipMemsetAsync(d_sdt, 0, HW * ts, h->stream); hipMemsetAsync(d_iy, 0, HW * 2, h->stream); }   // probe build: leave the x pass as the last DT launch (its stamps are then readable)
  else launch_dt_pass(d_tasks + nx, (int)tasks.size() - nx, d_maps, nullptr, nullptr, h->d_biasw, budget, tsz, nt, 0, h->stream);
  std::vector<int16_t> hx(HW), hy(HW);
  HIPCHK(h, hipMemcpyAsync(out, d_sdt, HW * ts, hipMemcpyDeviceToHost, h->stream));   // the y pass's scores, untouched
  HIPCHK(h, hipMemcpyAsync(hx.data(), d_ixT, HW * 2, hipMemcpyDeviceToHost, h->stream));   // the passes' own pointers
  HIPCHK(h, hipMemcpyAsync(hy.data(), d_iy, HW * 2, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < HW; ++i) {   // pointer composition of compute() (:233-244), or the true arg-max one
    const int m_ = (int)(i / cols), n_ = (int)(i - (size_t)m_ * cols);
    int x, y;
    if (!h->opt.dt_correct_ptr) { x = hx[i]; y = hy[(size_t)m_ * cols + x]; }
    else { y = hy[i]; x = hx[(size_t)y * cols + n_]; }
    if (ix) ix[i] = x;
    if (iy) iy[i] = y;
  }
  hipFree(d_in); hipFree(d_tmp); hipFree(d_sdt); hipFree(d_ixT); hipFree(d_iy);
  hipFree(d_maps); hipFree(d_tasks);
  return PBD_OK;
}
int pbd_dt2d(pbd_handle* h, const float* in, int rows, int cols, double ax, double bx, double ay, double by, int osx,
             int osy, float* out, int32_t* ix, int32_t* iy) {
  return dt2d_(h, in, rows, cols, ax, bx, ay, by, osx, osy, out, ix, iy, 4);
}
int pbd_dt2d_f64(pbd_handle* h, const double* in, int rows, int cols, double ax, double bx, double ay, double by, int osx,
                 int osy, double* out, int32_t* ix, int32_t* iy) {
  return dt2d_(h, in, rows, cols, ax, bx, ay, by, osx, osy, out, ix, iy, 8);
}

static int hog_u8_(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride, void* out, int* cell_w, int* cell_h, int ts) {
  if (!h || !im || w < 3 || hgt < 3 || (cn != 1 && cn != 3) || stride < w * cn) return PBD_ERR_ARG;
  CHECK_SCALAR(h, ts);
  ON_DEVICE(h);
  const int sbin = h->md.sbin;
  LevelDev L{};
  L.iw = w; L.ih = hgt;
  L.bw = (int)std::round((float)w / (float)sbin); L.bh = (int)std::round((float)hgt / (float)sbin);
  L.cw = std::max(L.bw - 2, 0); L.ch = std::max(L.bh - 2, 0);
  if (cell_w) *cell_w = L.cw;
  if (cell_h) *cell_h = L.ch;
  if (L.cw == 0 || L.ch == 0) return PBD_OK;
  int tc = 16;
  while (tc > 2 && hog_lds_bytes(sbin, tc, ts) > 150 * 1024) tc /= 2;
  std::vector<HogTile> tiles;
  for (int y = 0; y < L.ch; y += tc) for (int x = 0; x < L.cw; x += tc) tiles.push_back(HogTile{0, y, x, 0});
  uint8_t* d_im; char* d_feat; LevelDev* d_lv; HogTile* d_tiles;
  HIPCHK(h, hipMalloc(&d_im, (size_t)w * hgt * cn)); HIPCHK(h, hipMalloc(&d_feat, (size_t)L.cw * L.ch * PBD_FLEN * ts));
  HIPCHK(h, hipMalloc(&d_lv, sizeof(L))); HIPCHK(h, hipMalloc(&d_tiles, sizeof(HogTile) * tiles.size()));
  HIPCHK(h, hipMemcpy2D(d_im, (size_t)w * cn, im, stride, (size_t)w * cn, hgt, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(d_lv, &L, sizeof(L), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(d_tiles, tiles.data(), sizeof(HogTile) * tiles.size(), hipMemcpyHostToDevice));
  launch_hog(d_tiles, (int)tiles.size(), d_lv, d_im, d_feat, ts, cn, sbin, tc, h->d_hog_lut, nullptr, 0, PBD_DEPTH_8U, h->stream);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, d_feat, (size_t)L.cw * L.ch * PBD_FLEN * ts, hipMemcpyDeviceToHost));
  hipFree(d_im); hipFree(d_feat); hipFree(d_lv); hipFree(d_tiles);
  return PBD_OK;
}
int pbd_hog_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride, float* out, int* cell_w, int* cell_h) {
  return hog_u8_(h, im, w, hgt, cn, stride, out, cell_w, cell_h, 4);
}
int pbd_hog_u8_f64(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride, double* out, int* cell_w, int* cell_h) {
  return hog_u8_(h, im, w, hgt, cn, stride, out, cell_w, cell_h, 8);
}

int pbd_resize_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride, uint8_t* out, int ow, int oh) {
  if (!h || !im || !out || w <= 0 || hgt <= 0 || ow <= 0 || oh <= 0 || (cn != 1 && cn != 3) || stride < w * cn) return PBD_ERR_ARG;
  ON_DEVICE(h);
  uint8_t *d_im, *d_out;
  HIPCHK(h, hipMalloc(&d_im, (size_t)w * hgt * cn)); HIPCHK(h, hipMalloc(&d_out, (size_t)ow * oh * cn));
  HIPCHK(h, hipMemcpy2D(d_im, (size_t)w * cn, im, stride, (size_t)w * cn, hgt, hipMemcpyHostToDevice));
  PyrJob job{0, 0, w, hgt, ow, oh}, *d_job;
  HIPCHK(h, hipMalloc(&d_job, sizeof(job)));
  HIPCHK(h, hipMemcpy(d_job, &job, sizeof(job), hipMemcpyHostToDevice));
  launch_resize(d_job, 1, ow * oh, cn, w * cn, d_im, d_out, h->stream);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, d_out, (size_t)ow * oh * cn, hipMemcpyDeviceToHost));
  hipFree(d_im); hipFree(d_out); hipFree(d_job);
  return PBD_OK;
}

int pbd_pyrdown_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride, uint8_t* out) {
  if (!h || !im || !out || w <= 0 || hgt <= 0 || (cn != 1 && cn != 3) || stride < w * cn) return PBD_ERR_ARG;
  ON_DEVICE(h);
  const size_t sb = (size_t)w * hgt * cn, db = (size_t)((w + 1) / 2) * ((hgt + 1) / 2) * cn;
  uint8_t* d_buf;
  HIPCHK(h, hipMalloc(&d_buf, sb + db));
  HIPCHK(h, hipMemcpy2D(d_buf, (size_t)w * cn, im, stride, (size_t)w * cn, hgt, hipMemcpyHostToDevice));
  PyrJob job{0, (unsigned long long)sb, w, hgt, (w + 1) / 2, (hgt + 1) / 2}, *d_job;
  HIPCHK(h, hipMalloc(&d_job, sizeof(job)));
  HIPCHK(h, hipMemcpy(d_job, &job, sizeof(job), hipMemcpyHostToDevice));
  launch_pyrdown(d_job, 1, job.dw, job.dh, cn, d_buf, h->stream);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, d_buf + sb, db, hipMemcpyDeviceToHost));
  hipFree(d_buf); hipFree(d_job);
  return PBD_OK;
}

int pbd_nms_map(pbd_handle* h, const float* src, int rows, int cols, int sz, uint8_t* dst) {
  if (!h || !src || !dst || rows <= 0 || cols <= 0 || sz < 0) return PBD_ERR_ARG;
  ON_DEVICE(h);
  float* d_src; uint8_t* d_dst;
  const size_t n = (size_t)rows * cols;
  HIPCHK(h, hipMalloc(&d_src, n * 4)); HIPCHK(h, hipMalloc(&d_dst, n));
  HIPCHK(h, hipMemcpy(d_src, src, n * 4, hipMemcpyHostToDevice));
  launch_nms_map(d_src, rows, cols, sz, d_dst, h->stream);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(dst, d_dst, n, hipMemcpyDeviceToHost));
  hipFree(d_src); hipFree(d_dst);
  return PBD_OK;
}

// ---- host-side post-processing (include/Candidate.hpp:91-99, 277-304) --------
int pbd_candidates_sort(pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int count, int mp) {
  if (!heads || count < 0 || mp <= 0) return PBD_ERR_ARG;
  std::vector<int> order(count);
  for (int i = 0; i < count; ++i) order[i] = i;
  // Candidate::descending; stable, so equal scores keep detect() order (std::sort leaves it unspecified)
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return heads[a].score > heads[b].score; });
  std::vector<pbd_candidate_head> h2(count);
  std::vector<int32_t> b2(boxes ? (size_t)count * mp * 4 : 0), l2(locs ? (size_t)count * mp * 3 : 0);
  for (int i = 0; i < count; ++i) {
    h2[i] = heads[order[i]];
    if (boxes) memcpy(&b2[(size_t)i * mp * 4], boxes + (size_t)order[i] * mp * 4, sizeof(int32_t) * mp * 4);
    if (locs) memcpy(&l2[(size_t)i * mp * 3], locs + (size_t)order[i] * mp * 3, sizeof(int32_t) * mp * 3);
  }
  if (count) memcpy(heads, h2.data(), sizeof(pbd_candidate_head) * count);
  if (boxes && count) memcpy(boxes, b2.data(), b2.size() * sizeof(int32_t));
  if (locs && count) memcpy(locs, l2.data(), l2.size() * sizeof(int32_t));
  return PBD_OK;
}

int pbd_candidates_nms(pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int count, int mp, int im_w, int im_h,
                       float overlap, int* kept) {
  if (!heads || !boxes || !kept || count < 0 || mp <= 0 || im_w <= 0 || im_h <= 0) return PBD_ERR_ARG;
  std::vector<uint8_t> scratch((size_t)im_w * im_h, 0);
  int keep = 0;
  for (int n = 0; n < count; ++n) {
    const int32_t* b = boxes + (size_t)n * mp * 4;
    int x = b[0], y = b[1], bw = b[2], bh = b[3];  // Candidate::boundingBox(): union of the part rects
    for (int p = 0; p < heads[n].nparts; ++p) {
      const int32_t* q = b + p * 4;
      const int x1 = std::min(x, q[0]), y1 = std::min(y, q[1]);
      bw = std::max(x + bw, q[0] + q[2]) - x1;
      bh = std::max(y + bh, q[1] + q[3]) - y1;
      x = x1; y = y1;
    }
    int ix1 = std::max(x, 0), iy1 = std::max(y, 0);  // & bounds
    int iw = std::min(x + bw, im_w) - ix1, ih = std::min(y + bh, im_h) - iy1;
    if (iw <= 0 || ih <= 0) ix1 = iy1 = iw = ih = 0;
    double sum = 0;
    for (int yy = iy1; yy < iy1 + ih; ++yy)
      for (int xx = ix1; xx < ix1 + iw; ++xx) sum += scratch[(size_t)yy * im_w + xx];
    if (sum / (double)(iw * ih) > (double)overlap) continue;  // :296
    for (int yy = iy1; yy < iy1 + ih; ++yy) memset(&scratch[(size_t)yy * im_w + ix1], 1, iw);
    if (keep != n) {
      heads[keep] = heads[n];
      memmove(boxes + (size_t)keep * mp * 4, boxes + (size_t)n * mp * 4, sizeof(int32_t) * mp * 4);
      if (locs) memmove(locs + (size_t)keep * mp * 3, locs + (size_t)n * mp * 3, sizeof(int32_t) * mp * 3);
    }
    keep++;
  }
  *kept = keep;
  return PBD_OK;
}

// ---- instrumentation ---------------------------------------------------------
int pbd_get_stage_ms(const pbd_handle* h, float ms[6]) {
  if (!h || !ms) return PBD_ERR_ARG;
  for (int i = 0; i < 6; ++i) ms[i] = h->stage_ms[i];
  return PBD_OK;
}
int pbd_set_profiling(pbd_handle* h, int on) {
  if (!h) return PBD_ERR_ARG;
  h->profiling = on != 0;
  return PBD_OK;
}
int pbd_get_work(const pbd_handle* h, double work[6]) {
  if (!h || !work) return PBD_ERR_ARG;
  if (h->fw == 0) return PBD_ERR_STATE;
  const pbd_model_desc& m = h->md;
  double C = 0, pix = 0;
  for (int l = 0; l < h->nlevels; ++l) {
    if (!h->lv[l].active) continue;
    C += (double)h->lv[l].cw * h->lv[l].ch;
    pix += (double)h->lv[l].iw * h->lv[l].ih * h->fcn * h->fesz;   // (bytes: pixels of the frame's own depth)
  }
  // reference element types (scores of type T: 4 or 8 bytes, int32 pointers)
  const double ts = h->ts;
  double per_cell = 0, dtmaps = 0;
  for (int c = 0; c < m.ncomponents; ++c) {
    const int p0 = h->part_offset[c], cnp = h->part_offset[c + 1] - p0;
    for (int p = 1; p < cnp; ++p) {
      const double K = h->parts[p0 + p].K, L = h->parts[p0 + h->parts[p0 + p].parent].K;
      per_cell += ts * K + 12 * L + 2 * ts * L;
      dtmaps += K;
    }
    per_cell += ts * h->parts[p0].K + ts + 4;
  }
  work[0] = pix + 32.0 * ts * C;
  work[1] = 32.0 * ts * C + ts * m.nfilters * C + ts * m.nfilters * m.kh * m.kw * m.flen;
  work[2] = 2.0 * C * m.nfilters * m.kh * m.kw * m.flen;
  work[3] = C * per_cell;
  work[4] = C;
  work[5] = C * dtmaps;
  return PBD_OK;
}
#ifdef PBD_PROBES
#define PROBE_RC PBD_OK
#else
#define PROBE_RC PBD_ERR_UNSUPPORTED   /* stamps exist only in libpbd_hip_probes.so (make probes) */
#endif
#ifdef PBD_PROBES
extern "C" int pbd_debug_dt_trace(unsigned long long* t, unsigned* hw, int* nlaunch) { return dt_debug_trace(t, hw, nlaunch); }
#endif
int pbd_debug_dt_stamps(unsigned long long* out) { if (!out) return PBD_ERR_ARG; dt_debug_read(out); return PROBE_RC; }
int pbd_debug_hog_stamps(unsigned long long* out) { if (!out) return PBD_ERR_ARG; hog_debug_read(out); return PROBE_RC; }
int pbd_debug_conv_stamps(unsigned long long* out) { if (!out) return PBD_ERR_ARG; conv_debug_read(out); return PROBE_RC; }

int pbd_dp_timer(pbd_handle* h, int reset, double* avg_ms, int* nframes) {
  if (!h) return PBD_ERR_ARG;
  if (avg_ms) *avg_ms = h->dp_frames ? h->dp_ms_sum / h->dp_frames : 0.0;
  if (nframes) *nframes = h->dp_frames;
  if (reset) { h->dp_ms_sum = 0; h->dp_frames = 0; }
  return PBD_OK;
}

}  // extern "C"
#pragma GCC visibility pop
