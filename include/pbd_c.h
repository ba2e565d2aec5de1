/*
 * pbd_c.h — C ABI of libpbd_hip.so: the MI355X (gfx950) inference path behind
 * PartsBasedDetector<float>::detect().
 *
 * Every entry point below names the reference interface it replaces
 * (file:line under wg-perception/PartsBasedDetector).  The ABI is plain C:
 * opaque handle, POD structs, raw pointers + sizes, int status codes; no C++
 * exception ever crosses it.  INTEGRATION.md shows the reference-side
 * adaptors (IFeatures / IConvolutionEngine / DynamicProgram / detect()) that
 * bind it.
 *
 * Conventions
 *   - all matrices are dense row-major, "H x W" = rows x cols;
 *   - feature maps are cell-major with `flen` floats contiguous per cell
 *     (the reference's H x (W*flen) cv::Mat, src/HOGFeatures.cpp:178);
 *   - filters are kh x (kw*flen) floats, same interleave
 *     (src/MatlabIOModel.cpp:106-125);
 *   - a handle owns one GPU + one stream; calls on one handle must be
 *     serialised by the caller (the reference detector is not re-entrant
 *     either: src/HOGFeatures.cpp:99,106-107).
 */
#ifndef PBD_C_H_
#define PBD_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (reference: assert / CV_Error / bool) */
enum {
  PBD_OK = 0,
  PBD_ERR_ARG = 1,         /* bad pointer / size / index                       */
  PBD_ERR_UNSUPPORTED = 2, /* CV_StsUnsupportedFormat, src/HOGFeatures.cpp:141 */
  PBD_ERR_CAPACITY = 3,    /* output array too small; *count = needed          */
  PBD_ERR_HIP = 4,         /* HIP runtime failure (see pbd_last_error)         */
  PBD_ERR_STATE = 5,       /* stage called before its producer stage           */
  PBD_ERR_RCCL = 6         /* RCCL failure in a pbd_group gather (see pbd_group_last_error) */
};

/* ---- model: POD mirror of include/Model.hpp:49-122 ------------------------
 * Parts of all components are stored back to back ("flat part" index fp);
 * mixtures of all parts are stored back to back ("flat mixture" index fm).
 *   component c owns flat parts  part_offset[c] .. part_offset[c+1]-1
 *   flat part fp owns mixtures   mix_offset[fp] .. mix_offset[fp+1]-1
 * filterid/defid/biasid are 0-based like the C++ Model after zeroIndex
 * (src/MatlabIOModel.cpp:155-166).  For a child part with K mixtures whose
 * parent has L mixtures, bias(mm)[m] = biasw[biasid[fm0+mm] + m]
 * (include/Parts.hpp:172-175), deformation defw[defid[fm0+mm]][0..3] and
 * anchor anchors[defid[fm0+mm]] (include/Parts.hpp:179-183).
 */
typedef struct pbd_model_desc {
  int32_t nfilters;      /* Model::filters().size()                            */
  int32_t kh, kw;        /* filter rows / cols, 1..9, UNIFORM over the bank: the reference allows a size per filter
                            (include/Parts.hpp:185-187) but every model its tools write is uniform; a mixed bank is
                            not representable here (the C++ adaptor refuses it with PBD_ERR_UNSUPPORTED)             */
  int32_t flen;          /* Model::flen()  (32)                                */
  int32_t norient;       /* Model::norient() (18)                              */
  int32_t sbin;          /* Model::binsize()                                   */
  int32_t interval;      /* Model::nscales() (levels per octave)               */
  float thresh;          /* Model::thresh()                                    */
  const float* filters;  /* [nfilters][kh][kw*flen]                            */
  int32_t ndefs;
  const float* defw;     /* [ndefs][4]  = {wxx, wx, wyy, wy}                   */
  const int32_t* anchors;/* [ndefs][2]  = {x, y}, 0-based                      */
  int32_t nbias;
  const float* biasw;    /* [nbias]                                            */
  int32_t ncomponents;
  const int32_t* part_offset; /* [ncomponents+1]                               */
  const int32_t* parentid;    /* [nparts_total], index local to the component  */
  const int32_t* mix_offset;  /* [nparts_total+1]                              */
  const int32_t* filterid;    /* [nmix_total]                                  */
  const int32_t* defid;       /* [nmix_total] (root: ignored)                  */
  const int32_t* biasid;      /* [nmix_total]                                  */
} pbd_model_desc;

/* ---- options -------------------------------------------------------------- */
enum {
  PBD_CONV_AUTO = 0,  /* banks of fewer than 16 filters: EXACT.  From 16 filters on, any kh x kw: float handles take SPLIT (round 5:
                         ABI version 4; rounds 3-4: MFMA), double handles MFMA.  Neither is bit-identical to the reference's
                         summation order (|delta| <= 2e-5 on HOG features, north_star 1e-4): callers who need the reference's
                         bits ask for PBD_CONV_EXACT.  pbd_get_conv_mode() tells what a handle resolved to.               */
  PBD_CONV_EXACT = 1, /* VALU direct correlation, reference summation order:
                         bit-identical to src/filter.cpp:3899-3922 + pdf+=pdfc */
  PBD_CONV_MFMA = 2,  /* MFMA implicit GEMM (k-ordered fma chain) for any kh x kw: fp32
                         v_mfma_f32_16x16x4_f32 for float handles, fp64
                         v_mfma_f64_16x16x4_f64 for double handles             */
  PBD_CONV_SPLIT = 3, /* float handles, any kh x kw (x 32 channels): the fp32 products on the bf16 matrix units through EXACT
                         three-way splits (x = h + m + l, three bfloat16 of 8 significant bits; the six partial products above
                         2^-24 relative on v_mfma_f32_32x32x16_bf16, fp32 accumulators): fp32 in, fp32 out, errors of the size of
                         PBD_CONV_MFMA's, on hardware the vector ALU does not share.  Weights must be finite
                         and below 3e38 in magnitude (bfloat16's range) — checked by pbd_create —, and so must features
                         handed in through pbd_set_level_features (HOG features are <= 0.4): an out-of-domain or non-finite
                         feature is refused there with PBD_ERR_ARG (PBD_CONV_MFMA / PBD_CONV_EXACT carry such values as
                         ordinary fp32)                                                                          */
  PBD_CONV_SPLIT_F16 = 4 /* opt-in, never what AUTO resolves to.  float handles: TWO binary16 parts per operand (11 significant
                         bits each, operands scaled by powers of two into binary16's range: features by 2^12, a bank's weights to
                         max |w| 2^e in [2^13, 2^14)) and the THREE products above 2^-22 relative on v_mfma_f32_32x32x16_f16, fp32
                         accumulators, responses scaled back exactly — half the matrix instructions of PBD_CONV_SPLIT.  Operands
                         are carried to 23 of their 24 bits; measured errors against fp64 on HOG features: those of
                         PBD_CONV_SPLIT.  Domain: |feature| < 16 (HOG features are <= 1; features handed in
                         through pbd_set_level_features must respect it), weights finite; a feature below 2^-26 or a
                         weight below 2^-27 max |w| loses relative (not absolute) precision.                      */
};
/* Scalar type T of the instantiation (src/PartsBasedDetector.cpp:132-133):
 * PartsBasedDetector<float> (src/demo.cpp:85) or PartsBasedDetector<double>
 * (ros/Node.hpp:121, cells/detect.cpp:93).  Features, responses, scores and
 * the distance transform are computed and stored in T; model weights stay
 * float and are widened where the reference widens them; candidates carry
 * float scores for both (include/Candidate.hpp:72).                          */
enum { PBD_SCALAR_F32 = 0, PBD_SCALAR_F64 = 1 };
/* Depth of an input image: the values of cv::Mat::depth() the reference dispatches on (src/HOGFeatures.cpp:136-146:
 * CV_8U = 0, CV_16U = 2, CV_32F = 5, CV_64F = 6; anything else: CV_Error(StsUnsupportedFormat) -> PBD_ERR_UNSUPPORTED).
 * The *_u8 entry points are the 8-bit case; pbd_detect_image / pbd_pyramid_image take any of the four.             */
enum { PBD_DEPTH_8U = 0, PBD_DEPTH_16U = 2, PBD_DEPTH_32F = 5, PBD_DEPTH_64F = 6 };
typedef struct pbd_options {
  int32_t device;        /* HIP device ordinal                                 */
  int32_t conv_mode;     /* PBD_CONV_*                                         */
  int32_t max_candidates;/* device-side candidate capacity per frame           */
  int32_t dt_correct_ptr;/* 0 = reference pointer composition
                            (include/DistanceTransform.hpp:233-244), 1 = true
                            arg-max composition                                */
  int32_t level_begin;   /* process pyramid levels [level_begin, level_end)    */
  int32_t level_end;     /* <=0: all levels (multi-GPU level sharding)         */
  int32_t scalar_type;   /* PBD_SCALAR_F32 (default) or PBD_SCALAR_F64; a double
                            handle answers the *_f64 stage entry points
                            instead of the float ones                          */
  int32_t graph;         /* 1: capture the ~40 launches of a frame into a hipGraph once per frame geometry and
                            replay it (one hipGraphLaunch per frame instead of ~40 launches); 0: eager launches */
  int32_t reserved[2];   /* [0]: nms_sz — 0 (default, the reference's state: its call site is commented out,
                                 src/PartsBasedDetector.cpp:86): no suppression; sz > 0: nonMaximaSuppression(rootv, sz)
                                 (src/nms.cpp:84-129) of every (level, component) root-score plane ON THE DEVICE between
                                 min() and argmin(): only roots that are above the threshold AND the strict maximum of
                                 their (2 sz + 1)^2 neighbourhood (block rule of nms.cpp) are back-tracked and returned
                                 (ABI version 4; versions <= 3 ignored the slot);
                            [1]: dp_mode — 0: a part's messages are folded by its own x pass wherever the model allows it
                                 (no filter id shared inside a component, <= 8 mixtures per part, <= 8 children per part),
                                 1: the three-kernel structure (x pass, y pass, reduce + accumulated planes) for every model,
                                 2: fold + the compact memory plan (stage buffers that are never live together share
                                    memory; automatic for large frames, e.g. 1920x1080: 1.47 GB instead of 3.3 GB per
                                    handle): after min() / detect() the image, feature and response getters answer
                                    PBD_ERR_STATE                                                                       */
} pbd_options;
/* The layout of pbd_options and pbd_model_desc is frozen from PBD_ABI_VERSION 3 on: new options take a reserved slot
 * or a new entry point, fields are never inserted.  pbd_abi_version() returns the version the LIBRARY was built with;
 * a binding compares it with the header it was compiled against (round 2 inserted `graph` in front of reserved[],
 * which nothing could detect).                                                                                       */
#define PBD_ABI_VERSION 4
int pbd_abi_version(void);
/* Version history: 3 = rounds 3-4.  4 (round 5) = PBD_CONV_AUTO resolves to PBD_CONV_SPLIT for float handles (numerics of
 * AUTO change in the last bits: rounds 3-4 resolved to PBD_CONV_MFMA, and before that to EXACT for banks other than 5 x 5),
 * PBD_CONV_SPLIT, PBD_CONV_SPLIT_F16, pbd_detect_image / pbd_pyramid_image / pbd_get_level_image_raw (PBD_DEPTH_*), pbd_tune_plan, pbd_options.reserved[0] = nms_sz, pbd_get_conv_mode, pbd_get_stage_state, pbd_group_comm_size.  Struct layouts unchanged.
 * Round 6 keeps version 4 (no entry point, layout or result changed); refinements of existing entries: pbd_set_level_features refuses
 * features outside a split bank's domain (PBD_ERR_ARG), pbd_tune_plan drops the handle's plan on return, pbd_detect_image replays a
 * hipGraph under pbd_options.graph.                                                                                             */

/* ---- output record: include/Candidate.hpp:56-111 --------------------------
 * One candidate = head + max_parts boxes (x, y, width, height as cv::Rect)
 * + max_parts part locations (x, y, mixture) in cells of its pyramid level.
 * max_parts = pbd_max_parts(handle).  Part confidences are the reference's:
 * root = rootv, every other part 0.0 (src/DynamicProgram.cpp:241-244).
 */
typedef struct pbd_candidate_head {
  float score;        /* Candidate::score()                                    */
  int32_t component;  /* Candidate::component()                                */
  int32_t level;      /* pyramid level the root was found at                   */
  int32_t nparts;     /* parts of that component                               */
} pbd_candidate_head;

typedef struct pbd_handle pbd_handle;

/* Restrict the handle to an arbitrary SET of pyramid levels (n = 0: all levels again), intersected with
 * [level_begin, level_end).  Levels never interact (src/DynamicProgram.cpp:83-87 loops over (level,
 * component) pairs independently), so one large frame shards across GPUs by level with no data-path
 * collective: every rank rebuilds the (cheap) image pyramid and runs HOG / pdf / min / argmin on its own
 * cost-balanced level set (SURVEY 8e, configs[3]); the union of the ranks' candidates is the frame's.   */
int pbd_set_levels(pbd_handle* h, const int32_t* levels, int n);
/* PartsBasedDetector<T>::distributeModel (src/PartsBasedDetector.cpp:102-127)
 * incl. SpatialConvolutionEngine::setFilters (src/SpatialConvolutionEngine.cpp:133-159)
 * and Parts construction (include/Parts.hpp:229-235).                        */
int pbd_create(const pbd_model_desc* model, const pbd_options* opt, pbd_handle** out);
int pbd_destroy(pbd_handle* h);
const char* pbd_last_error(const pbd_handle* h);
int pbd_max_parts(const pbd_handle* h);
/* set the stream all work of this handle is enqueued on (hipStream_t as void*) */
int pbd_set_stream(pbd_handle* h, void* hip_stream);

/* PartsBasedDetector<T>::detect(im, candidates) (src/PartsBasedDetector.cpp:69-95).
 * `im` is a host pointer to an 8-bit image, cn = 1 or 3 (BGR interleaved),
 * stride in bytes.  Candidates are written in the order of a single-threaded
 * reference run (level, component, row-major root location); at most
 * `capacity`; *count = number found (PBD_ERR_CAPACITY if > capacity).
 * heads[capacity], boxes[capacity][max_parts][4], locs[capacity][max_parts][3]
 * (boxes / locs may be NULL).                                                 */
int pbd_detect_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride,
                  pbd_candidate_head* heads, int32_t* boxes, int32_t* locs,
                  int capacity, int* count);
/* Input depth.  The reference dispatches features<uint8_t|uint16_t|float|double> on im.depth()
 * (src/HOGFeatures.cpp:136-146); its three callers (src/demo.cpp:90, ros/Node.cpp:183,
 * cells/detect.cpp:224) all pass CV_8U BGR, which is what the *_u8 entry points (device-resident images, batches,
 * graph replay, groups) are built and tuned for.  pbd_detect_image takes a host image of any of the four depths
 * (`depth` = PBD_DEPTH_*, stride in BYTES, a multiple of the element size): pyramid levels in the image's own type —
 * cv::resize interpolating in floating point with float coefficients, cv::pyrDown as FltCast<T, 8> (ushort: the
 * integer form), restated from OpenCV 2.4 like the 8-bit pair and equally unpinned (no reference test holds any
 * pyramid value) —, gradients in the pixel type's promoted arithmetic, everything from the histograms on unchanged.
 * Single host frames (replayed as a hipGraph under pbd_options.graph like 8-bit plans, round 6); batches, device-resident entry points
 * and groups stay 8-bit — by design: no caller of the reference hands over anything else, and every further entry point is a surface
 * to test against an oracle that is itself unpinned for these depths.  PBD_DEPTH_8U forwards to pbd_detect_u8.  Any other depth:
 * PBD_ERR_UNSUPPORTED, the counterpart of CV_Error(StsUnsupportedFormat) (:141-145).                               */
int pbd_detect_image(pbd_handle* h, const void* im, int depth, int w, int hgt, int cn, int stride,
                     pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* count);
/* same, image already resident in device memory (tightly packed or strided)  */
int pbd_detect_dev_u8(pbd_handle* h, const void* d_im, int w, int hgt, int cn, int stride,
                      pbd_candidate_head* heads, int32_t* boxes, int32_t* locs,
                      int capacity, int* count);
/* asynchronous halves of pbd_detect_dev_u8: enqueue all kernels + the D2H of
 * the candidate buffer on the handle's stream; collect after the stream (or
 * the caller's event) has completed.  Lets a caller overlap frames.          */
int pbd_detect_enqueue_dev_u8(pbd_handle* h, const void* d_im, int w, int hgt, int cn, int stride);
int pbd_detect_collect(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs,
                       int capacity, int* count);
/* asynchronous pbd_detect_u8: the H2D copy of the host image is enqueued on the handle's stream in
 * front of the kernels (truly asynchronous when `im` is pinned: hipHostMalloc / hipHostRegister; a pageable
 * image is staged by the runtime).  `im` must stay valid until pbd_detect_collect returns.               */
int pbd_detect_enqueue_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride);

/* ---- a batch of same-sized frames on ONE handle (BASELINE configs[2] gives every GPU 4 frames) ----------
 * The frames of a batch go through every stage TOGETHER: one launch (or one chain of launches) per stage for the whole
 * batch — the same kernels with `nframes` times the blocks per launch, which fills the chip in the thin rounds of the DP
 * and pays every launch tail once per batch.  Results per frame are identical to pbd_detect_u8.
 * Frame f's candidates land at heads[f*capacity], boxes[f*capacity*max_parts*4], locs[f*capacity*max_parts*3] (boxes /
 * locs may be NULL), counts[f] = number found; PBD_ERR_CAPACITY if a frame exceeds `capacity` or the batch exceeds
 * pbd_options.max_candidates.  1 <= nframes <= 64; the work tables are re-planned when nframes (or the size) changes.
 * _enqueue_dev_: the frames already in device memory, tightly packed, back to back; collect after either enqueue.   */
int pbd_detect_batch_u8(pbd_handle* h, const uint8_t* const* ims, int nframes, int w, int hgt, int cn, int stride,
                        pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* counts);
int pbd_detect_batch_enqueue_u8(pbd_handle* h, const uint8_t* const* ims, int nframes, int w, int hgt, int cn, int stride);
int pbd_detect_batch_enqueue_dev_u8(pbd_handle* h, const void* d_ims, int nframes, int w, int hgt, int cn);
int pbd_detect_batch_collect(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* counts);

/* ---- one process, several GPUs (SURVEY 8b "Threading", 8e) ------------------
 * The reference's hosts are single processes (src/demo.cpp:85-103, ros/Node.cpp:183, cells/detect.cpp:224).
 * A pbd_group owns one handle per listed device (a device may be listed more than once: several frames in
 * flight on it) and drives them from the calling thread; frames and pyramid levels never interact
 * (src/DynamicProgram.cpp:83-87), so there is no data-path collective, only the gather of the members'
 * candidate buffers:
 *   PBD_GATHER_RCCL  ncclAllGather over the members' devices (RCCL over xGMI; librccl is loaded at run time) of a
 *                    fixed-size block {count, first records}, then ONE D2H on member 0; members holding more
 *                    records than the block hand the remainder over directly;
 *   PBD_GATHER_HOST  one small D2H per member + concatenation on the host (same result; used when librccl is
 *                    missing or a device is listed twice — RCCL wants distinct devices);
 *   PBD_GATHER_AUTO  RCCL when possible, else host.
 * Results are identical to running the frames one after the other on a single handle.                      */
enum { PBD_GATHER_AUTO = 0, PBD_GATHER_HOST = 1, PBD_GATHER_RCCL = 2 };
typedef struct pbd_group pbd_group;
/* opt->device is ignored (devices[] decides); every other option applies to all members                     */
int pbd_group_create(const pbd_model_desc* model, const pbd_options* opt, const int32_t* devices, int ndevices,
                     int gather_mode, pbd_group** out);
int pbd_group_destroy(pbd_group* g);
const char* pbd_group_last_error(const pbd_group* g);
int pbd_group_size(const pbd_group* g);
int pbd_group_gather_mode(const pbd_group* g);          /* PBD_GATHER_HOST or PBD_GATHER_RCCL actually in use   */
int pbd_group_comm_size(const pbd_group* g);            /* ranks of the RCCL communicator (ncclCommCount); 0 = host gather (ABI 4) */
pbd_handle* pbd_group_member(pbd_group* g, int i);     /* borrowed: stage entry points, pbd_get_stage_ms, ...   */
/* BASELINE configs[2]: a batch of same-sized frames, frame f on member f % size, all members busy at once.
 * Frame f's candidates land at heads[f*capacity], boxes[f*capacity*max_parts*4], locs[f*capacity*max_parts*3]
 * (boxes / locs may be NULL), counts[f] = number found; PBD_ERR_CAPACITY if any frame exceeds `capacity`.   */
int pbd_group_detect_batch_u8(pbd_group* g, const uint8_t* const* ims, int nframes, int w, int hgt, int cn, int stride,
                              pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* counts);
/* BASELINE configs[3]: ONE frame, its pyramid levels spread over the members by greedy LPT on the cell counts
 * (every member rebuilds the cheap image pyramid); output in the order a single handle produces.            */
int pbd_group_detect_u8(pbd_group* g, const uint8_t* im, int w, int hgt, int cn, int stride,
                        pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* count);

/* ---- stage entry points (parity testing; same kernels as detect) ----------
 * IFeatures::nscales/scales (include/IFeatures.hpp:54-63) + pyramid geometry
 * of HOGFeatures<T>::pyramid (src/HOGFeatures.cpp:98-127).  Arrays sized
 * >= *nlevels (call with NULL arrays to query).  img_* = level image size,
 * cell_* = feature map size, scales = IFeatures::scales().                   */
int pbd_pyramid_geometry(const pbd_handle* h, int w, int hgt, int* nlevels,
                         int32_t* img_w, int32_t* img_h, int32_t* cell_w, int32_t* cell_h,
                         float* scales);
/* HOGFeatures<T>::pyramid (src/HOGFeatures.cpp:95-151): image pyramid + HOG  */
int pbd_pyramid_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride);
int pbd_get_level_image(pbd_handle* h, int level, uint8_t* out /* img_h*img_w*cn */);
/* HOGFeatures<T>::pyramid for an image of any accepted depth (pbd_detect_image), and a level image in the frame's own pixel
 * type (out_bytes >= img_h * img_w * cn * element size)                                                              */
int pbd_pyramid_image(pbd_handle* h, const void* im, int depth, int w, int hgt, int cn, int stride);
int pbd_get_level_image_raw(pbd_handle* h, int level, void* out, size_t out_bytes);
int pbd_get_level_features(pbd_handle* h, int level, float* out /* cell_h*cell_w*flen */);
/* (a handle on a split-product bank refuses features outside the bank's domain — PBD_CONV_SPLIT: finite, |f| < 3e38;
 *  PBD_CONV_SPLIT_F16: |f| < 16 — with PBD_ERR_ARG; nothing is uploaded then)                                          */
int pbd_set_level_features(pbd_handle* h, int level, const float* in);
int pbd_get_level_features_f64(pbd_handle* h, int level, double* out);
int pbd_set_level_features_f64(pbd_handle* h, int level, const double* in);
/* declare a frame geometry without running the pyramid (inject features)     */
int pbd_begin_frame(pbd_handle* h, int w, int hgt, int cn);
/* SpatialConvolutionEngine::pdf (src/SpatialConvolutionEngine.cpp:106-124)   */
int pbd_pdf(pbd_handle* h);
int pbd_get_level_response(pbd_handle* h, int level, int filter, float* out /* cell_h*cell_w */);
int pbd_set_level_response(pbd_handle* h, int level, int filter, const float* in);
int pbd_get_level_response_f64(pbd_handle* h, int level, int filter, double* out);
int pbd_set_level_response_f64(pbd_handle* h, int level, int filter, const double* in);
/* DynamicProgram<T>::min (src/DynamicProgram.cpp:66-173)                      */
int pbd_dp_min(pbd_handle* h);
/* Ix/Iy/Ik[level][component][part][parent mixture] as int32 cell_h*cell_w     */
int pbd_get_dp_pointers(pbd_handle* h, int level, int component, int part, int parent_mix,
                        int32_t* ix, int32_t* iy, int32_t* ik);
int pbd_get_root(pbd_handle* h, int level, int component, float* rootv, int32_t* rooti);
int pbd_get_root_f64(pbd_handle* h, int level, int component, double* rootv, int32_t* rooti);
/* DynamicProgram<T>::argmin takes rootv / rooti / Ix / Iy / Ik as arguments (include/DynamicProgram.hpp:75).  A caller
 * whose tables are not the ones this handle's min() left on the device (another engine's min(), edited tables) hands
 * them over here before pbd_dp_argmin; the next pbd_dp_min / detect goes back to the handle's own tables.  The first
 * pbd_set_dp_pointers after a min() materialises all composed planes once (the planes not handed in keep min()'s).  */
int pbd_set_root(pbd_handle* h, int level, int component, const float* rootv, const int32_t* rooti);
int pbd_set_root_f64(pbd_handle* h, int level, int component, const double* rootv, const int32_t* rooti);
int pbd_set_dp_pointers(pbd_handle* h, int level, int component, int part, int parent_mix,
                        const int32_t* ix, const int32_t* iy, const int32_t* ik);
/* Which stage buffers of the current frame plan hold valid data: state[0] level images, [1] features, [2] responses,
 * [3] the DP tables (what the getters / the next stage would answer PBD_ERR_STATE for when 0).  On a handle with the
 * compact memory plan (reserved[1] = 2, or automatic for large frames) min() reuses the image / feature memory and
 * transforms the responses in place: [0..2] read 0 afterwards, and a response / feature setter makes its stage valid
 * again only once EVERY plane of the active levels has been handed in.  A caller that caches what is resident on the
 * device (host/pbd_host.hpp: content fingerprints) must drop that knowledge when a flag reads 0.                      */
int pbd_get_stage_state(const pbd_handle* h, int32_t state[4]);
/* The filter bank this handle runs (PBD_CONV_EXACT / _MFMA / _SPLIT): what PBD_CONV_AUTO resolved to at pbd_create
 * (SpatialConvolutionEngine is the reference's only engine, src/PartsBasedDetector.cpp:111; the choice here is numerical:
 * see the enum).  Negative: error code.                                                                              */
int pbd_get_conv_mode(const pbd_handle* h);
/* DynamicProgram<T>::argmin (src/DynamicProgram.cpp:189-255).  With tables handed in and NO min() of this handle on the
 * frame, every pointer table and every root table of the handle's levels must have been provided (PBD_ERR_STATE else). */
int pbd_dp_argmin(pbd_handle* h, pbd_candidate_head* heads, int32_t* boxes, int32_t* locs,
                  int capacity, int* count);

/* ---- stand-alone primitives ------------------------------------------------
 * DistanceTransform<T>::compute (include/DistanceTransform.hpp:202-245) with
 * Quadratic(ax,bx), Quadratic(ay,by) and anchor (osx, osy); host arrays.      */
int pbd_dt2d(pbd_handle* h, const float* in, int rows, int cols,
             double ax, double bx, double ay, double by, int osx, int osy,
             float* out, int32_t* ix, int32_t* iy);
int pbd_dt2d_f64(pbd_handle* h, const double* in, int rows, int cols,
                 double ax, double bx, double ay, double by, int osx, int osy,
                 double* out, int32_t* ix, int32_t* iy);   /* DistanceTransform<double> */
/* HOGFeatures<T>::features<uint8_t> (src/HOGFeatures.cpp:168-341), one image */
int pbd_hog_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride,
               float* out, int* cell_w, int* cell_h);
int pbd_hog_u8_f64(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride,
                   double* out, int* cell_w, int* cell_h);  /* HOGFeatures<double> */
/* cv::resize(INTER_LINEAR) / cv::pyrDown on 8-bit images as used at
 * src/HOGFeatures.cpp:116,122 (this library's definition)     */
int pbd_resize_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride,
                  uint8_t* out, int ow, int oh);
int pbd_pyrdown_u8(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride,
                   uint8_t* out);
/* Neubeck-Van Gool block NMS on a score map (src/nms.cpp:84-129)             */
int pbd_nms_map(pbd_handle* h, const float* src, int rows, int cols, int sz, uint8_t* dst);

/* ---- host-side post-processing (include/Candidate.hpp:91-99, 277-304) ------
 * operate in place on the arrays returned by detect; pure host code.         */
int pbd_candidates_sort(pbd_candidate_head* heads, int32_t* boxes, int32_t* locs,
                        int count, int max_parts);
int pbd_candidates_nms(pbd_candidate_head* heads, int32_t* boxes, int32_t* locs,
                       int count, int max_parts, int im_w, int im_h, float overlap, int* kept);

/* ---- instrumentation --------------------------------------------------------
 * GPU time (ms, hipEvent) of the stages of the last synchronous detect:
 * [0] image pyramid [1] HOG [2] pdf [3] dp min [4] argmin [5] total            */
int pbd_get_stage_ms(const pbd_handle* h, float ms[6]);
/* enable per-stage events (adds host syncs between stages; off by default)   */
int pbd_set_profiling(pbd_handle* h, int on);
/* algorithmic bytes / flops of the last frame geometry (SURVEY §8d formulas):
 * [0] B_hog [1] B_pdf [2] F_pdf [3] B_dp [4] cells [5] dt_elements            */
int pbd_get_work(const pbd_handle* h, double work[6]);
/* device memory held by the handle: the buffers and work tables of the current frame geometry (everything a
 * re-plan frees) and the model-sized allocations made at create.  Either pointer may be NULL.                     */
int pbd_get_footprint(const pbd_handle* h, size_t* frame_bytes, size_t* model_bytes);
/* Measure the planner's distance-transform block geometry on the caller's own frames instead of trusting its rule (float handles:
 * 256 lanes / 40 KB against 128 lanes / 25 KB per block; results are bit-identical under either): `batch` copies of the host image
 * per call (1 = single frames, the reference's call shape; > 1 = pbd_detect_batch_u8), 2 warm-up + 3 timed calls per geometry, the
 * dp_min stage's GPU time.  The faster one is kept for every later plan of this handle; *chosen = 1 / 2 (0: nothing to choose —
 * double handles), ms[0..1] = the two medians.  im == NULL: back to the rule.  Synchronous; costs ten calls.  The handle's plan is
 * dropped on return (as after pbd_set_levels): stage getters answer PBD_ERR_STATE until the next frame; pbd_get_stage_ms keeps the
 * caller's last figures.  Refused (PBD_ERR_STATE) on a member of an RCCL-gathering pbd_group and while a frame is pending.          */
int pbd_tune_plan(pbd_handle* h, const uint8_t* im, int w, int hgt, int cn, int stride, int batch, int* chosen, double ms[2]);
/* average GPU ms of the DP-min kernels alone over frames since the last reset
 * (HIP events on the handle's stream around the DP stage)                    */
int pbd_dp_timer(pbd_handle* h, int reset, double* avg_ms, int* nframes);
/* The pbd_debug_* entry points below report something only in the probe build of the library
 * (make -C partsbaseddetector_amd/csrc probes -> libpbd_hip_probes.so, -DPBD_PROBES: per-phase stamps inside
 * the kernels + environment tuning knobs); the product library compiles neither and returns
 * PBD_ERR_UNSUPPORTED.
 * debug: 100 MHz wall-clock stamps of block 0 of the last distance-transform launch at its six
 * phase boundaries (setup, line load, envelope scan, read-out, pointer store, end)            */
int pbd_debug_dt_stamps(unsigned long long out[8]);
/* same for the HOG kernel: tile staging, gradient, histogram, energy+normalisers, features */
int pbd_debug_hog_stamps(unsigned long long out[8]);
/* same for the MFMA filter bank: tile staging, K loop, barrier, epilogue                    */
int pbd_debug_conv_stamps(unsigned long long out[8]);

#ifdef __cplusplus
}
#endif
#endif /* PBD_C_H_ */
