// pbd_group.cpp — one process, several GPUs (include/pbd_c.h "pbd_group").
//
// The reference's hosts are single processes (src/demo.cpp:85-103, ros/Node.cpp:183, cells/detect.cpp:224); this is
// the entry a maintainer binds there to use every GPU of a node: one handle per device, all driven asynchronously
// from the calling thread (every handle has its own stream; nothing blocks until the gather).  Frames and pyramid
// levels never interact (src/DynamicProgram.cpp:83-87), so the only exchange is the gather of the members'
// candidate blocks: ncclAllGather (RCCL over xGMI, librccl loaded with dlopen so that the library has no
// load-time dependency on it) followed by ONE D2H on member 0, or one small D2H per member (host concatenation).
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <new>
#include <set>
#include "pbd_internal.hpp"

namespace {
// the slice of rccl.h this file uses (ncclResult_t / ncclDataType_t are plain enums; ncclInt8 = 0)
typedef struct ncclComm* ncclComm_t;
typedef int (*fnCommInitAll)(ncclComm_t*, int, const int*);
typedef int (*fnCommDestroy)(ncclComm_t);
typedef int (*fnAllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
typedef int (*fnGroup)(void);
typedef const char* (*fnErrStr)(int);
typedef int (*fnCommCount)(const ncclComm_t, int*);
struct Rccl {
  void* so = nullptr;
  fnCommInitAll CommInitAll = nullptr;
  fnCommDestroy CommDestroy = nullptr;
  fnAllGather AllGather = nullptr;
  fnGroup GroupStart = nullptr, GroupEnd = nullptr;
  fnErrStr GetErrorString = nullptr;
  fnCommCount CommCount = nullptr;     // optional (pbd_group_comm_size)
  bool load() {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (so) break;
    }
    if (!so) return false;
    CommInitAll = (fnCommInitAll)dlsym(so, "ncclCommInitAll");
    CommDestroy = (fnCommDestroy)dlsym(so, "ncclCommDestroy");
    AllGather = (fnAllGather)dlsym(so, "ncclAllGather");
    GroupStart = (fnGroup)dlsym(so, "ncclGroupStart");
    GroupEnd = (fnGroup)dlsym(so, "ncclGroupEnd");
    GetErrorString = (fnErrStr)dlsym(so, "ncclGetErrorString");
    CommCount = (fnCommCount)dlsym(so, "ncclCommCount");
    return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString;
  }
};
}  // namespace

struct pbd_group {
  std::vector<pbd_handle*> m;
  std::vector<int> dev;
  int mode = PBD_GATHER_HOST;
  std::string err;
  Rccl rccl;
  std::vector<ncclComm_t> comms;
  std::vector<char*> d_send, d_recv;   // per member: its block / every member's block
  char* h_recv = nullptr;              // pinned, size() blocks (filled from member 0)
  size_t block = 0;                    // bytes of one {count, pad, first records} block
  int shard_w = 0, shard_h = 0, shard_cn = 0;   // geometry the members' level sets were computed for (0: all levels)
  // scratch of one gather
  std::vector<int> found;
  std::vector<std::vector<char>> extra;   // records beyond the block (rare), per member
};

static int gfail(pbd_group* g, int code, const std::string& msg) {
  if (g) g->err = msg;
  return code;
}
#define GHIP(g, call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) return gfail(g, PBD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define GNCCL(g, call)                                                             \
  do {                                                                             \
    int r_ = (call);                                                               \
    if (r_ != 0) return gfail(g, PBD_ERR_RCCL, std::string(#call) + ": " + (g)->rccl.GetErrorString(r_)); \
  } while (0)
#define GMEMBER(g, i, call)                                                        \
  do {                                                                             \
    int r_ = (call);                                                               \
    if (r_ != PBD_OK) return gfail(g, r_, "member " + std::to_string(i) + ": " + pbd_last_error((g)->m[i])); \
  } while (0)

// all members have enqueued a frame (or nothing: `active[i]` false) — bring every active member's records to the host.
// After it, found[i] and rec_ptrs(i) are valid.
static int gather(pbd_group* g, const std::vector<char>& active) {
  const int n = (int)g->m.size();
  g->found.assign(n, 0);
  g->extra.assign(n, {});
  if (g->mode == PBD_GATHER_RCCL) {
    for (int i = 0; i < n; ++i) {
      GHIP(g, hipSetDevice(g->dev[i]));
      if (!active[i]) GHIP(g, hipMemsetAsync(g->d_send[i], 0, 16, g->m[i]->stream));   // count = 0
    }
    GNCCL(g, g->rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
      GHIP(g, hipSetDevice(g->dev[i]));
      GNCCL(g, g->rccl.AllGather(g->d_send[i], g->d_recv[i], g->block, /*ncclInt8*/ 0, g->comms[i], g->m[i]->stream));
    }
    GNCCL(g, g->rccl.GroupEnd());
    GHIP(g, hipSetDevice(g->dev[0]));
    GHIP(g, hipMemcpyAsync(g->h_recv, g->d_recv[0], g->block * n, hipMemcpyDeviceToHost, g->m[0]->stream));
    GHIP(g, hipStreamSynchronize(g->m[0]->stream));
    for (int i = 0; i < n; ++i) {
      if (!active[i]) continue;
      pbd_handle* h = g->m[i];
      GHIP(g, hipSetDevice(g->dev[i]));
      GHIP(g, hipStreamSynchronize(h->stream));   // the all-gather on this member's stream has completed as well
      const int found = *(const int*)(g->h_recv + g->block * i);
      g->found[i] = found;
      GMEMBER(g, i, pbd_i_finish_frame(h, found));   // also fetches records beyond the block into h->h_cand_out
    }
  } else {
    for (int i = 0; i < n; ++i) {
      if (!active[i]) continue;
      pbd_handle* h = g->m[i];
      GHIP(g, hipSetDevice(g->dev[i]));
      GHIP(g, hipStreamSynchronize(h->stream));
      g->found[i] = h->h_cand_count[0];
      GMEMBER(g, i, pbd_i_finish_frame(h, g->found[i]));
    }
  }
  return PBD_OK;
}
// record j of member i after gather()
static const char* rec_ptr(const pbd_group* g, int i, int j) {
  const pbd_handle* h = g->m[i];
  if (g->mode == PBD_GATHER_RCCL && j < PBD_FIRST_COPY) return g->h_recv + g->block * i + 16 + h->cand_stride * j;
  return h->h_cand_out + h->cand_stride * j;   // host mode, or the remainder pbd_i_finish_frame fetched
}

// An error in the middle of a batch or a gather (a member over its candidate capacity, a failed enqueue, an RCCL
// error) must not leave the OTHER members with a frame in flight: they would answer every later call with "previous
// frame not collected", and in RCCL mode a member cannot be collected on its own.  Every failing group call ends
// here: wait for whatever the members still have enqueued and drop it; the first error stays in g->err.
static void drain(pbd_group* g) {
  for (size_t i = 0; i < g->m.size(); ++i) {
    pbd_handle* h = g->m[i];
    if (!h->pending) continue;
    hipSetDevice(g->dev[i]);
    hipStreamSynchronize(h->stream);
    h->pending = false;
  }
  (void)hipGetLastError();
}
static int batch_impl(pbd_group* g, const uint8_t* const* ims, int nframes, int w, int hgt, int cn, int stride,
                      pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* counts);
static int frame_impl(pbd_group* g, const uint8_t* im, int w, int hgt, int cn, int stride, pbd_candidate_head* heads,
                      int32_t* boxes, int32_t* locs, int capacity, int* count);

#pragma GCC visibility push(default)
extern "C" {

int pbd_group_create(const pbd_model_desc* model, const pbd_options* opt, const int32_t* devices, int ndev, int gather_mode,
                     pbd_group** out) {
  if (!out) return PBD_ERR_ARG;
  *out = nullptr;
  pbd_group* g = new (std::nothrow) pbd_group();
  if (!g) return PBD_ERR_ARG;
  *out = g;   // returned even on failure so that pbd_group_last_error() can be read; destroy it either way
  if (!devices || ndev <= 0 || ndev > 64) return gfail(g, PBD_ERR_ARG, "devices: 1..64 device ordinals");
  if (gather_mode != PBD_GATHER_AUTO && gather_mode != PBD_GATHER_HOST && gather_mode != PBD_GATHER_RCCL)
    return gfail(g, PBD_ERR_ARG, "gather_mode: PBD_GATHER_AUTO / _HOST / _RCCL");
  pbd_options o{};
  if (opt) o = *opt;
  for (int i = 0; i < ndev; ++i) {
    o.device = devices[i];
    pbd_handle* h = nullptr;
    int rc = pbd_create(model, &o, &h);
    if (rc != PBD_OK) {
      std::string msg = "member " + std::to_string(i) + " (device " + std::to_string(devices[i]) + "): " + pbd_last_error(h);
      if (h) pbd_destroy(h);
      return gfail(g, rc, msg);
    }
    g->m.push_back(h);
    g->dev.push_back(devices[i]);
  }
  const bool distinct = std::set<int>(g->dev.begin(), g->dev.end()).size() == g->dev.size();
  g->mode = PBD_GATHER_HOST;
  if (gather_mode != PBD_GATHER_HOST) {
    std::string why;
    if (!distinct) why = "a device is listed more than once (RCCL wants one rank per device)";
    else if (!g->rccl.load()) why = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "missing symbols");
    if (why.empty()) {
      g->comms.assign(ndev, nullptr);
      int r = g->rccl.CommInitAll(g->comms.data(), ndev, g->dev.data());
      if (r != 0) { why = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r); g->comms.clear(); }
    }
    if (why.empty()) {
      pbd_handle* h0 = g->m[0];
      const int first = std::min(PBD_FIRST_COPY, h0->opt.max_candidates);
      g->block = 16 + h0->cand_stride * first;
      g->d_send.assign(ndev, nullptr); g->d_recv.assign(ndev, nullptr);
      for (int i = 0; i < ndev; ++i) {
        GHIP(g, hipSetDevice(g->dev[i]));
        GHIP(g, hipMalloc((void**)&g->d_send[i], g->block));
        GHIP(g, hipMalloc((void**)&g->d_recv[i], g->block * ndev));
        GHIP(g, hipMemset(g->d_send[i], 0, g->block));
        g->m[i]->d_gsend = g->d_send[i];
      }
      GHIP(g, hipSetDevice(g->dev[0]));
      GHIP(g, hipHostMalloc((void**)&g->h_recv, g->block * ndev));
      g->mode = PBD_GATHER_RCCL;
    } else if (gather_mode == PBD_GATHER_RCCL) {
      return gfail(g, PBD_ERR_RCCL, "PBD_GATHER_RCCL requested but " + why);
    }
  }
  return PBD_OK;
}

int pbd_group_destroy(pbd_group* g) {
  if (!g) return PBD_ERR_ARG;
  for (size_t i = 0; i < g->m.size(); ++i) {
    hipSetDevice(g->dev[i]);
    if (g->m[i]->stream) hipStreamSynchronize(g->m[i]->stream);
  }
  for (ncclComm_t c : g->comms) if (c) g->rccl.CommDestroy(c);
  for (size_t i = 0; i < g->m.size(); ++i) {
    hipSetDevice(g->dev[i]);
    if (i < g->d_send.size() && g->d_send[i]) hipFree(g->d_send[i]);
    if (i < g->d_recv.size() && g->d_recv[i]) hipFree(g->d_recv[i]);
    g->m[i]->d_gsend = nullptr;
    pbd_destroy(g->m[i]);
  }
  if (g->h_recv) hipHostFree(g->h_recv);
  if (g->rccl.so) dlclose(g->rccl.so);
  delete g;
  return PBD_OK;
}

const char* pbd_group_last_error(const pbd_group* g) { return g ? g->err.c_str() : "null group"; }
int pbd_group_size(const pbd_group* g) { return g ? (int)g->m.size() : 0; }
int pbd_group_gather_mode(const pbd_group* g) { return g ? g->mode : PBD_GATHER_HOST; }
// ranks of the RCCL communicator the gather runs on, as RCCL reports it (ncclCommCount of member 0's communicator); 0: host gather
int pbd_group_comm_size(const pbd_group* g) {
  if (!g || g->mode != PBD_GATHER_RCCL || g->comms.empty() || !g->comms[0] || !g->rccl.CommCount) return 0;
  int n = 0;
  return g->rccl.CommCount(g->comms[0], &n) == 0 ? n : -1;
}
pbd_handle* pbd_group_member(pbd_group* g, int i) { return (g && i >= 0 && i < (int)g->m.size()) ? g->m[i] : nullptr; }

static int all_levels(pbd_group* g) {   // undo a level sharding left behind by pbd_group_detect_u8
  if (!g->shard_w) return PBD_OK;
  for (size_t i = 0; i < g->m.size(); ++i) {
    hipSetDevice(g->dev[i]);
    GMEMBER(g, i, pbd_set_levels(g->m[i], nullptr, 0));
  }
  g->shard_w = g->shard_h = g->shard_cn = 0;
  return PBD_OK;
}

int pbd_group_detect_batch_u8(pbd_group* g, const uint8_t* const* ims, int nframes, int w, int hgt, int cn, int stride,
                              pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* counts) {
  if (!g || !ims || nframes < 0 || !heads || !counts || capacity < 0) return PBD_ERR_ARG;
  const int rc = batch_impl(g, ims, nframes, w, hgt, cn, stride, heads, boxes, locs, capacity, counts);
  if (rc != PBD_OK) drain(g);
  return rc;
}

int pbd_group_detect_u8(pbd_group* g, const uint8_t* im, int w, int hgt, int cn, int stride, pbd_candidate_head* heads,
                        int32_t* boxes, int32_t* locs, int capacity, int* count) {
  if (!g || !im || !heads || capacity < 0) return PBD_ERR_ARG;
  const int rc = frame_impl(g, im, w, hgt, cn, stride, heads, boxes, locs, capacity, count);
  if (rc != PBD_OK) drain(g);
  return rc;
}

}  // extern "C"
#pragma GCC visibility pop

static int batch_impl(pbd_group* g, const uint8_t* const* ims, int nframes, int w, int hgt, int cn, int stride,
                      pbd_candidate_head* heads, int32_t* boxes, int32_t* locs, int capacity, int* counts) {
  const int n = (int)g->m.size();
  int rc = all_levels(g);
  if (rc) return rc;
  const int mp = g->m[0]->max_parts;
  int status = PBD_OK;
  auto emit_frame = [&](int i, int f) -> int {      // member i's records -> the caller's slot of frame f
    counts[f] = g->found[i];
    std::vector<const char*> recs((size_t)g->found[i]);
    for (int j = 0; j < g->found[i]; ++j) recs[j] = rec_ptr(g, i, j);
    int r = pbd_i_emit(g->m[i], recs, heads + (size_t)f * capacity, boxes ? boxes + (size_t)f * capacity * mp * 4 : nullptr,
                       locs ? locs + (size_t)f * capacity * mp * 3 : nullptr, capacity);
    if (r == PBD_ERR_CAPACITY) { status = gfail(g, r, "frame " + std::to_string(f) + ": output capacity too small"); return PBD_OK; }
    if (r) return gfail(g, r, pbd_last_error(g->m[i]));
    return PBD_OK;
  };
  if (g->mode == PBD_GATHER_HOST) {
    // software pipeline: a member gets its next frame as soon as its previous one has been collected, so every
    // member always has a frame in flight (list a device several times to keep several frames in flight on it)
    g->found.assign(n, 0);
    std::vector<int> frame_of(n, -1);
    auto collect_member = [&](int i) -> int {
      pbd_handle* h = g->m[i];
      GHIP(g, hipSetDevice(g->dev[i]));
      GHIP(g, hipStreamSynchronize(h->stream));
      g->found[i] = h->h_cand_count[0];
      GMEMBER(g, i, pbd_i_finish_frame(h, g->found[i]));
      const int f = frame_of[i];
      frame_of[i] = -1;
      return emit_frame(i, f);
    };
    for (int f = 0; f < nframes; ++f) {
      const int i = f % n;
      if (!ims[f]) return gfail(g, PBD_ERR_ARG, "null frame pointer");
      if (frame_of[i] >= 0 && (rc = collect_member(i))) return rc;
      GMEMBER(g, i, pbd_detect_enqueue_u8(g->m[i], ims[f], w, hgt, cn, stride));
      frame_of[i] = f;
    }
    for (int f = std::max(0, nframes - n); f < nframes; ++f)      // drain in frame order
      if (frame_of[f % n] == f && (rc = collect_member(f % n))) return rc;
    return status;
  }
  for (int f0 = 0; f0 < nframes; f0 += n) {        // RCCL gather is a collective over the members: waves of n frames
    const int k = std::min(n, nframes - f0);
    std::vector<char> active(n, 0);
    for (int i = 0; i < k; ++i) {   // frame f0+i on member i: asynchronous H2D + all kernels, nothing waits here
      if (!ims[f0 + i]) return gfail(g, PBD_ERR_ARG, "null frame pointer");
      GMEMBER(g, i, pbd_detect_enqueue_u8(g->m[i], ims[f0 + i], w, hgt, cn, stride));
      active[i] = 1;
    }
    if ((rc = gather(g, active))) return rc;
    for (int i = 0; i < k; ++i)
      if ((rc = emit_frame(i, f0 + i))) return rc;
  }
  return status;
}

static int frame_impl(pbd_group* g, const uint8_t* im, int w, int hgt, int cn, int stride, pbd_candidate_head* heads,
                      int32_t* boxes, int32_t* locs, int capacity, int* count) {
  const int n = (int)g->m.size();
  int rc;
  if (g->shard_w != w || g->shard_h != hgt || g->shard_cn != cn) {
    // greedy LPT over the levels' cell counts (the cost of every stage is proportional to them): levels in decreasing
    // cost, each to the least loaded member; 1920x1080 on 8 members: makespan = level 0 = 13 % of the cells
    int nl = 0;
    std::vector<int32_t> cw(PBD_MAX_LEVELS), ch(PBD_MAX_LEVELS);
    rc = pbd_pyramid_geometry(g->m[0], w, hgt, &nl, nullptr, nullptr, cw.data(), ch.data(), nullptr);
    if (rc) return gfail(g, rc, "image too small for the pyramid (src/HOGFeatures.cpp:99,114)");
    std::vector<int> order(nl);
    for (int l = 0; l < nl; ++l) order[l] = l;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (long long)cw[a] * ch[a] > (long long)cw[b] * ch[b]; });
    std::vector<double> load(n, 0.0);
    std::vector<std::vector<int32_t>> sets(n);
    for (int l : order) {
      int best = 0;
      for (int i = 1; i < n; ++i) if (load[i] < load[best]) best = i;
      sets[best].push_back(l);
      load[best] += (double)cw[l] * ch[l];
    }
    for (int i = 0; i < n; ++i) {
      hipSetDevice(g->dev[i]);
      if (sets[i].empty()) sets[i].push_back(nl - 1);   // more members than levels: a duplicate of the smallest level, dropped below
      GMEMBER(g, i, pbd_set_levels(g->m[i], sets[i].data(), (int)sets[i].size()));
    }
    g->shard_w = w; g->shard_h = hgt; g->shard_cn = cn;
  }
  std::vector<char> active(n, 1);
  for (int i = 0; i < n; ++i) GMEMBER(g, i, pbd_detect_enqueue_u8(g->m[i], im, w, hgt, cn, stride));
  if ((rc = gather(g, active))) return rc;
  std::vector<const char*> recs;
  std::set<std::vector<int>> seen;   // (level, comp, y, x): a level owned by two members (n > levels) counts once
  const int mp = g->m[0]->max_parts;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < g->found[i]; ++j) {
      const char* r = rec_ptr(g, i, j);
      const pbd_candidate_head* hd = (const pbd_candidate_head*)r;
      const int32_t* lc = (const int32_t*)(r + sizeof(pbd_candidate_head)) + (size_t)mp * 4;
      if (seen.insert({hd->level, hd->component, lc[1], lc[0]}).second) recs.push_back(r);
    }
  if (count) *count = (int)recs.size();
  rc = pbd_i_emit(g->m[0], recs, heads, boxes, locs, capacity);
  if (rc) return gfail(g, rc, pbd_last_error(g->m[0]));
  return PBD_OK;
}
