// pbd_host.hpp — C++ host side above the C ABI (include/pbd_c.h), mirroring the reference's
// interfaces so code written against wg-perception/PartsBasedDetector reads the same:
//
//   reference                                              here (namespace pbd)
//   IFeatures            include/IFeatures.hpp:49-73       pbd::IFeatures, pbd::HipHOGFeatures
//   IConvolutionEngine   include/IConvolutionEngine.hpp:44-68  pbd::IConvolutionEngine, pbd::HipConvolutionEngine
//   DynamicProgram<T>    include/DynamicProgram.hpp:61-77  pbd::DynamicProgram<T>
//   PartsBasedDetector<T> include/PartsBasedDetector.hpp:152-175  pbd::PartsBasedDetector<T>
//   Candidate            include/Candidate.hpp:56-111,277-304  pbd::Candidate
//   Model                include/Model.hpp:49-122          pbd::Model (+ pbd::BinaryModel reader)
//
// The reference passes cv::Mat; OpenCV is not a dependency of this library, so a minimal dense
// matrix (pbd::Mat: rows, cols, channels, 8U/32S/32F) carries the same data.  The OpenCV-typed
// adaptors for dropping the engines into the reference tree itself are in INTEGRATION.md.
// All numerics run in libpbd_hip.so; this header only marshals.  T = float or double, the two
// instantiations the reference declares (src/PartsBasedDetector.cpp:132-133).
#ifndef PBD_HOST_HPP_
#define PBD_HOST_HPP_

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/pbd_c.h"

namespace pbd {

enum { PBD_8U = 0, PBD_16U = 2, PBD_32S = 4, PBD_32F = 5, PBD_64F = 6 };  // depth codes (numerically OpenCV's CV_8U/16U/32S/32F/64F = pbd_c.h's PBD_DEPTH_*)
template <typename T> struct DataType;               // cv::DataType<T>::type
template <> struct DataType<float> { enum { type = PBD_32F, scalar = PBD_SCALAR_F32 }; };
template <> struct DataType<double> { enum { type = PBD_64F, scalar = PBD_SCALAR_F64 }; };

class Exception : public std::runtime_error {   // plays the role of cv::Exception
 public:
  int code;
  Exception(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

class Mat {
 public:
  int rows = 0, cols = 0;
  Mat() {}
  Mat(int r, int c, int depth, int cn = 1) { create(r, c, depth, cn); }
  void create(int r, int c, int depth, int cn = 1) {
    rows = r; cols = c; depth_ = depth; cn_ = cn;
    buf_.assign((size_t)r * c * cn * elem1(), 0);
  }
  bool empty() const { return buf_.empty(); }
  int depth() const { return depth_; }
  int channels() const { return cn_; }
  size_t elem1() const { return depth_ == PBD_8U ? 1 : depth_ == PBD_16U ? 2 : depth_ == PBD_64F ? 8 : 4; }
  size_t step() const { return (size_t)cols * cn_ * elem1(); }
  size_t bytes() const { return buf_.size(); }
  template <typename T> T* ptr(int r = 0) { return (T*)(buf_.data() + (size_t)r * step()); }
  template <typename T> const T* ptr(int r = 0) const { return (const T*)(buf_.data() + (size_t)r * step()); }
  template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
 private:
  int depth_ = PBD_8U, cn_ = 1;
  std::vector<uint8_t> buf_;
};

struct Rect { int x, y, width, height; };
struct Point { int x, y; };
typedef std::vector<int> vectori;
typedef std::vector<float> vectorf;
typedef std::vector<Mat> vectorMat;
typedef std::vector<vectorMat> vector2DMat;
typedef std::vector<vector2DMat> vector3DMat;
typedef std::vector<vector3DMat> vector4DMat;
typedef std::vector<vectori> vector2Di;
typedef std::vector<vector2Di> vector3Di;
typedef std::vector<vectorf> vector2Df;

// ---- include/Candidate.hpp:56-111 ---------------------------------------------------------
class Candidate {
  std::vector<Rect> parts_;
  vectorf confidence_;
  int component_ = 0;
 public:
  int level = -1;                        // extra: pyramid level of the root
  std::vector<int> locs;                 // extra: (x, y, mixture) per part, in cells
  const std::vector<Rect>& parts() const { return parts_; }
  const vectorf& confidence() const { return confidence_; }
  void addPart(Rect r, float c) { parts_.push_back(r); confidence_.push_back(c); }
  float score() const { return confidence_.empty() ? -std::numeric_limits<float>::infinity() : confidence_[0]; }
  void setScore(float c) { if (confidence_.empty()) confidence_.resize(1); confidence_[0] = c; }   // :76
  void setComponent(int c) { component_ = c; }
  int component() const { return component_; }
  void resize(const float factor) {      // :82-89: `int *= float` — converted to float, multiplied, truncated back
    for (Rect& r : parts_) {
      r.height = (int)((float)r.height * factor); r.width = (int)((float)r.width * factor);
      r.y = (int)((float)r.y * factor); r.x = (int)((float)r.x * factor);
    }
  }
  Rect boundingBox() const {             // union of the part rects (:103-109)
    Rect h = parts_[0];
    for (const Rect& q : parts_) {
      const int x1 = std::min(h.x, q.x), y1 = std::min(h.y, q.y);
      h.width = std::max(h.x + h.width, q.x + q.width) - x1;
      h.height = std::max(h.y + h.height, q.y + q.height) - y1;
      h.x = x1; h.y = y1;
    }
    return h;
  }
  static void sort(std::vector<Candidate>& c);                                         // :97-99
  static void nonMaximaSuppression(int im_w, int im_h, std::vector<Candidate>& c, float overlap = 0.0f);  // :277-304
};
typedef std::vector<Candidate> vectorCandidate;

// ---- include/Model.hpp:49-122 ---------------------------------------------------------------
class Model {
 protected:
  vectorMat filtersw_; vector2Df defw_; vectorf biasw_; std::vector<Point> anchors_;
  vector3Di biasid_, filterid_, defid_; vector2Di parentid_;
  std::string name_; int nscales_ = 0; float thresh_ = 0; int binsize_ = 0, flen_ = 0, norient_ = 0;
 public:
  virtual ~Model() {}
  vectorMat& filters() { return filtersw_; }
  vector2Df& def() { return defw_; }
  vectorf& bias() { return biasw_; }
  std::vector<Point>& anchors() { return anchors_; }
  vector3Di& filterid() { return filterid_; }
  vector3Di& biasid() { return biasid_; }
  vector3Di& defid() { return defid_; }
  vector2Di& parentid() { return parentid_; }
  std::string name() { return name_; }
  float thresh() const { return thresh_; }
  void setThresh(float t) { thresh_ = t; }
  int binsize() const { return binsize_; }
  int nscales() const { return nscales_; }
  int flen() const { return flen_; }
  int norient() const { return norient_; }
  int ncomponents() const { return (int)filterid_.size(); }
  virtual bool deserialize(const std::string& filename) = 0;
};

// Flat little-endian dump of the fields above (written by partsbaseddetector_amd.model.Model.save);
// the reference's own formats (cv::FileStorage XML/YAML, .mat) are SURVEY §8(f) "next".
class BinaryModel : public Model {
  static bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }
 public:
  bool deserialize(const std::string& filename) override {
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) return false;
    int32_t hd[12];
    char magic[8];
    bool ok = rd(f, magic, 8) && !memcmp(magic, "PBDMODL1", 8) && rd(f, hd, sizeof(hd));
    if (!ok) { fclose(f); return false; }
    const int nf = hd[0], kh = hd[1], kw = hd[2];
    flen_ = hd[3]; norient_ = hd[4]; binsize_ = hd[5]; nscales_ = hd[6];
    const int ndefs = hd[7], nbias = hd[8], ncomp = hd[9];
    // the header counts size every allocation below: bound them by what the file can actually hold
    fseek(f, 0, SEEK_END);
    const long long fsize = ftell(f);
    fseek(f, 8 + (long)sizeof(hd), SEEK_SET);
    const long long per_filter = (long long)kh * kw * flen_ * 4;
    if (nf <= 0 || kh <= 0 || kw <= 0 || kh > 64 || kw > 64 || flen_ <= 0 || flen_ > 1024 || ndefs < 0 || nbias <= 0 || ncomp <= 0 ||
        (long long)nf * per_filter > fsize || (long long)ndefs * 24 > fsize || (long long)nbias * 4 > fsize || (long long)ncomp * 4 > fsize) {
      fclose(f);
      return false;
    }
    ok = rd(f, &thresh_, 4);
    filtersw_.resize(nf);
    for (int n = 0; n < nf && ok; ++n) { filtersw_[n].create(kh, kw * flen_, PBD_32F); ok = rd(f, filtersw_[n].ptr<float>(), (size_t)kh * kw * flen_ * 4); }
    defw_.assign(ndefs, vectorf(4)); anchors_.resize(ndefs);
    for (int d = 0; d < ndefs && ok; ++d) ok = rd(f, defw_[d].data(), 16);
    for (int d = 0; d < ndefs && ok; ++d) { int32_t a[2]; ok = rd(f, a, 8); anchors_[d] = Point{a[0], a[1]}; }
    biasw_.resize(nbias);
    ok = ok && rd(f, biasw_.data(), (size_t)nbias * 4);
    filterid_.resize(ncomp); biasid_.resize(ncomp); defid_.resize(ncomp); parentid_.resize(ncomp);
    for (int c = 0; c < ncomp && ok; ++c) {
      int32_t np; ok = rd(f, &np, 4);
      if (!ok || np <= 0 || (long long)np * 8 > fsize) { ok = false; break; }
      filterid_[c].resize(np); biasid_[c].resize(np); defid_[c].resize(np); parentid_[c].resize(np);
      for (int p = 0; p < np && ok; ++p) {
        int32_t pk[2]; ok = rd(f, pk, 8);
        if (!ok || pk[1] <= 0 || (long long)pk[1] * 12 > fsize) { ok = false; break; }
        parentid_[c][p] = pk[0];
        filterid_[c][p].resize(pk[1]); biasid_[c][p].resize(pk[1]); defid_[c][p].resize(pk[1]);
        ok = ok && rd(f, filterid_[c][p].data(), 4 * pk[1]) && rd(f, defid_[c][p].data(), 4 * pk[1]) && rd(f, biasid_[c][p].data(), 4 * pk[1]);
      }
    }
    name_ = filename;
    fclose(f);
    return ok;
  }
  bool serialize(const std::string& filename) const {
    FILE* f = fopen(filename.c_str(), "wb");
    if (!f) return false;
    const int kh = filtersw_[0].rows, kw = filtersw_[0].cols / flen_;
    int32_t hd[12] = {(int32_t)filtersw_.size(), kh, kw, flen_, norient_, binsize_, nscales_, (int32_t)defw_.size(),
                      (int32_t)biasw_.size(), (int32_t)filterid_.size(), 0, 0};
    fwrite("PBDMODL1", 1, 8, f); fwrite(hd, 4, 12, f); fwrite(&thresh_, 4, 1, f);
    for (const Mat& m : filtersw_) fwrite(m.ptr<float>(), 4, (size_t)kh * kw * flen_, f);
    for (const vectorf& d : defw_) fwrite(d.data(), 4, 4, f);
    for (const Point& a : anchors_) { int32_t v[2] = {a.x, a.y}; fwrite(v, 4, 2, f); }
    fwrite(biasw_.data(), 4, biasw_.size(), f);
    for (size_t c = 0; c < filterid_.size(); ++c) {
      int32_t np = (int32_t)filterid_[c].size(); fwrite(&np, 4, 1, f);
      for (int p = 0; p < np; ++p) {
        const int32_t k = (int32_t)filterid_[c][p].size();
        int32_t pk[2] = {p ? parentid_[c][p] : -1, k}; fwrite(pk, 4, 2, f);
        std::vector<int32_t> d(k, 0), b(k, biasid_[c][p].empty() ? 0 : biasid_[c][p][0]);
        for (int q = 0; q < k && p > 0 && q < (int)defid_[c][p].size(); ++q) d[q] = defid_[c][p][q];
        for (int q = 0; q < k && q < (int)biasid_[c][p].size(); ++q) b[q] = biasid_[c][p][q];
        fwrite(filterid_[c][p].data(), 4, k, f); fwrite(d.data(), 4, k, f); fwrite(b.data(), 4, k, f);
      }
    }
    fclose(f);
    return true;
  }
  void assign(Model& o) {
    filtersw_ = o.filters(); defw_ = o.def(); biasw_ = o.bias(); anchors_ = o.anchors(); biasid_ = o.biasid();
    filterid_ = o.filterid(); defid_ = o.defid(); parentid_ = o.parentid(); name_ = o.name(); nscales_ = o.nscales();
    thresh_ = o.thresh(); binsize_ = o.binsize(); flen_ = o.flen(); norient_ = o.norient();
  }
};

// Content fingerprint of a host buffer (four independent multiply-add lanes over its 64-bit words).  The stage
// adaptors below take their inputs as ARGUMENTS, like the reference's interfaces do (IConvolutionEngine::pdf(features,
// ...), DynamicProgram::min(parts, scores, ...), argmin(..., rootv, rooti, ..., Ix, Iy, Ik, ...)): what the caller
// passes is what is processed.  The device keeps the fingerprint of everything it handed out; an argument whose
// bytes still match is already resident and is not uploaded again, anything else — another engine's output, a
// buffer edited in place — is uploaded first.  Nothing is ever answered from stale resident buffers.
static inline uint64_t fingerprint(const void* p, size_t bytes) {
  const uint8_t* b = (const uint8_t*)p;
  uint64_t a0 = 0x9E3779B97F4A7C15ull, a1 = 0xC2B2AE3D27D4EB4Full, a2 = 0x165667B19E3779F9ull, a3 = 0x27D4EB2F165667C5ull;
  size_t i = 0;
  for (; i + 32 <= bytes; i += 32) {
    uint64_t w[4];
    memcpy(w, b + i, 32);
    a0 = (a0 ^ w[0]) * 0x100000001B3ull + 1; a1 = (a1 ^ w[1]) * 0x100000001B3ull + 3;
    a2 = (a2 ^ w[2]) * 0x100000001B3ull + 5; a3 = (a3 ^ w[3]) * 0x100000001B3ull + 7;
  }
  for (; i < bytes; ++i) a0 = (a0 ^ b[i]) * 0x100000001B3ull + 11;
  return (a0 ^ (a1 << 1) ^ (a2 << 2) ^ (a3 << 3)) + bytes;
}

// ---- shared device handle -------------------------------------------------------------------
class Device {
 public:
  pbd_handle* h = nullptr;
  int scalar = PBD_SCALAR_F32;   // T of the detector that owns this device (cv::DataType<T>::type)
  int max_candidates = 4096;     // pbd_options.max_candidates: device-side capacity, also the host arrays' size
  // what the device holds, as fingerprints of the host copies handed out (0 = nothing resident / unknown)
  std::vector<uint64_t> fp_feat;                 // [level]
  std::vector<std::vector<uint64_t>> fp_resp;    // [level][filter]
  std::vector<std::vector<uint64_t>> fp_root;    // [level][component]: rootv ^ rooti
  std::vector<uint64_t> fp_tab;                  // per (level, component, part, parent mixture) in walk order: Ix ^ Iy ^ Ik
  bool tables_resident = false;                  // min() of THIS device produced the tables on the device
  std::vector<int32_t> cell_w, cell_h;           // level sizes of the current frame geometry
  vectorf scales;
  void geometry(int w, int hgt) {                // after pbd_pyramid_u8 / pbd_begin_frame
    int n = 0;
    check(pbd_pyramid_geometry(h, w, hgt, &n, 0, 0, 0, 0, 0));
    cell_w.resize(n); cell_h.resize(n); scales.resize(n);
    check(pbd_pyramid_geometry(h, w, hgt, &n, 0, 0, cell_w.data(), cell_h.data(), scales.data()));
    fp_feat.assign(n, 0); fp_resp.clear(); fp_root.clear(); fp_tab.clear(); tables_resident = false;
  }
  // What the fingerprints claim to be resident must still BE resident: on a handle with the compact memory plan min()
  // overwrites the features and the responses (pbd_get_stage_state), so an argument that matches a fingerprint taken
  // before that min() would be skipped and the DP would run on the overwritten planes.
  void dropStale() {
    int32_t st[4] = {1, 1, 1, 1};
    check(pbd_get_stage_state(h, st));
    if (!st[1]) std::fill(fp_feat.begin(), fp_feat.end(), 0);
    if (!st[2]) for (auto& v : fp_resp) std::fill(v.begin(), v.end(), 0);
    if (!st[3]) { tables_resident = false; fp_tab.clear(); for (auto& v : fp_root) std::fill(v.begin(), v.end(), 0); }
  }
  // a caller that brings features from another IFeatures declares the frame first (the level sizes follow from it)
  void beginFrame(int w, int hgt, int cn) { check(pbd_begin_frame(h, w, hgt, cn)); geometry(w, hgt); }
  void checkLevels(const vectorMat& v, int per_cell, const char* what) const {
    if (v.size() != cell_w.size()) throw Exception(PBD_ERR_STATE, std::string(what) + ": level count differs from the frame geometry of the device "
                                                                  "(run HipHOGFeatures::pyramid or Device::beginFrame for this image first)");
    for (size_t l = 0; l < v.size(); ++l)
      if (v[l].rows != cell_h[l] || v[l].cols != cell_w[l] * per_cell)
        throw Exception(PBD_ERR_ARG, std::string(what) + ": level " + std::to_string(l) + " has the wrong size for the frame geometry of the device");
  }
  std::vector<float> filters, defw, biasw;
  std::vector<int32_t> anchors, part_offset, parentid, mix_offset, filterid, defid, biasid;
  Device(Model& m, int device, int conv_mode, int scalar_type = PBD_SCALAR_F32, int max_cand = 4096) {
    pbd_model_desc d{};
    const int kh = m.filters()[0].rows, kw = m.filters()[0].cols / m.flen();
    // The reference's engine takes a size per filter (src/SpatialConvolutionEngine.cpp:133-159, include/Parts.hpp:185-187); the
    // C ABI describes a bank by ONE kh x kw (every model the reference's tools write is uniform: matlab/modelTransfer.m): a bank of
    // mixed sizes is refused here instead of being read with the first filter's size
    for (Mat& f : m.filters())
      if (f.rows != kh || f.cols != kw * m.flen())
        throw Exception(PBD_ERR_UNSUPPORTED, "distributeModel: filters of different sizes in one bank (pbd_model_desc carries one kh x kw)");
    for (Mat& f : m.filters()) filters.insert(filters.end(), f.ptr<float>(), f.ptr<float>() + (size_t)kh * kw * m.flen());
    for (vectorf& w : m.def()) defw.insert(defw.end(), w.begin(), w.begin() + 4);
    for (Point& a : m.anchors()) { anchors.push_back(a.x); anchors.push_back(a.y); }
    biasw = m.bias();
    part_offset.push_back(0); mix_offset.push_back(0);
    for (size_t c = 0; c < m.filterid().size(); ++c) {
      for (size_t p = 0; p < m.filterid()[c].size(); ++p) {
        parentid.push_back(p ? m.parentid()[c][p] : -1);
        for (size_t k = 0; k < m.filterid()[c][p].size(); ++k) {
          filterid.push_back(m.filterid()[c][p][k]);
          const vectori& dv = m.defid()[c][p]; const vectori& bv = m.biasid()[c][p];
          defid.push_back(p && k < dv.size() ? dv[k] : 0);
          biasid.push_back(bv.empty() ? 0 : bv[std::min(k, bv.size() - 1)]);
        }
        mix_offset.push_back((int32_t)filterid.size());
      }
      part_offset.push_back((int32_t)parentid.size());
    }
    d.nfilters = (int)m.filters().size(); d.kh = kh; d.kw = kw; d.flen = m.flen(); d.norient = m.norient();
    d.sbin = m.binsize(); d.interval = m.nscales(); d.thresh = m.thresh();
    d.filters = filters.data(); d.ndefs = (int)m.def().size(); d.defw = defw.data(); d.anchors = anchors.data();
    d.nbias = (int)biasw.size(); d.biasw = biasw.data(); d.ncomponents = (int)m.filterid().size();
    d.part_offset = part_offset.data(); d.parentid = parentid.data(); d.mix_offset = mix_offset.data();
    d.filterid = filterid.data(); d.defid = defid.data(); d.biasid = biasid.data();
    pbd_options opt{};
    opt.device = device; opt.conv_mode = conv_mode; opt.scalar_type = scalar_type; opt.max_candidates = max_cand;
    scalar = scalar_type; max_candidates = max_cand > 0 ? max_cand : 4096;
    if (pbd_abi_version() != PBD_ABI_VERSION) throw Exception(PBD_ERR_UNSUPPORTED, "libpbd_hip.so was built for another pbd_c.h (pbd_abi_version)");
    const int rc = pbd_create(&d, &opt, &h);
    if (rc != PBD_OK) { std::string msg = h ? pbd_last_error(h) : "pbd_create failed"; if (h) pbd_destroy(h); h = nullptr; throw Exception(rc, msg); }
  }
  ~Device() { if (h) pbd_destroy(h); }
  void check(int rc) const { if (rc != PBD_OK) throw Exception(rc, pbd_last_error(h)); }
};

// ---- include/IFeatures.hpp:49-73 --------------------------------------------------------------
class IFeatures {
 public:
  virtual ~IFeatures() {}
  virtual size_t binsize() const = 0;
  virtual size_t nscales() const = 0;
  virtual vectorf scales() const = 0;
  virtual void pyramid(const Mat& im, vectorMat& pyrafeatures) = 0;
};

class HipHOGFeatures : public IFeatures {          // include/HOGFeatures.hpp:52-88
  std::shared_ptr<Device> dev_; size_t binsize_, nscales_; vectorf scales_;
 public:
  HipHOGFeatures(std::shared_ptr<Device> d, size_t binsize, size_t nscales) : dev_(d), binsize_(binsize), nscales_(nscales) {}
  size_t binsize() const override { return binsize_; }
  size_t nscales() const override { return nscales_; }
  vectorf scales() const override { return scales_; }
  void pyramid(const Mat& im, vectorMat& pyrafeatures) override {   // src/HOGFeatures.cpp:95-151
    // :136-146 dispatches features<uint8_t | uint16_t | float | double> on im.depth(); any other depth: StsUnsupportedFormat (the library refuses it)
    dev_->check(pbd_pyramid_image(dev_->h, im.ptr<uint8_t>(), im.depth(), im.cols, im.rows, im.channels(), (int)im.step()));
    dev_->geometry(im.cols, im.rows);
    const int n = (int)dev_->cell_w.size();
    const std::vector<int32_t>&cw = dev_->cell_w, &ch = dev_->cell_h;
    scales_ = dev_->scales; nscales_ = n;
    pyrafeatures.clear(); pyrafeatures.resize(n);
    for (int l = 0; l < n; ++l) {
      const bool f64 = dev_->scalar == PBD_SCALAR_F64;   // HOGFeatures<T>: Mat of DataType<T>::type
      pyrafeatures[l].create(ch[l], cw[l] * 32, f64 ? PBD_64F : PBD_32F);
      if (ch[l] > 0 && cw[l] > 0)
        dev_->check(f64 ? pbd_get_level_features_f64(dev_->h, l, pyrafeatures[l].ptr<double>())
                        : pbd_get_level_features(dev_->h, l, pyrafeatures[l].ptr<float>()));
      dev_->fp_feat[l] = fingerprint(pyrafeatures[l].ptr<uint8_t>(), pyrafeatures[l].bytes());
    }
  }
};

// ---- include/IConvolutionEngine.hpp:44-68 -------------------------------------------------------
class IConvolutionEngine {
 public:
  virtual ~IConvolutionEngine() {}
  virtual void pdf(const vectorMat& features, vector2DMat& responses) = 0;
  virtual void setFilters(const vectorMat& filters) = 0;
};

class HipConvolutionEngine : public IConvolutionEngine {   // include/SpatialConvolutionEngine.hpp:44-58
  std::shared_ptr<Device> dev_; size_t nfilters_ = 0;
 public:
  explicit HipConvolutionEngine(std::shared_ptr<Device> d) : dev_(d) {}
  void setFilters(const vectorMat& filters) override { nfilters_ = filters.size(); }  // uploaded by pbd_create
  void pdf(const vectorMat& features, vector2DMat& responses) override {  // src/SpatialConvolutionEngine.cpp:106-124
    const bool f64 = dev_->scalar == PBD_SCALAR_F64;   // SpatialConvolutionEngine(type_)
    dev_->checkLevels(features, 32, "pdf(features)");
    // the features that are processed are the ones passed in: levels whose bytes the device does not already hold
    // (another IFeatures' pyramid, a pyramid edited after pyramid()) are uploaded first
    for (size_t l = 0; l < features.size(); ++l) {
      if (features[l].empty()) continue;
      if (features[l].depth() != (f64 ? PBD_64F : PBD_32F)) throw Exception(PBD_ERR_ARG, "pdf(features): element type differs from the detector's T");
      const uint64_t fp = fingerprint(features[l].ptr<uint8_t>(), features[l].bytes());
      if (fp == dev_->fp_feat[l]) continue;
      dev_->check(f64 ? pbd_set_level_features_f64(dev_->h, (int)l, features[l].ptr<double>())
                      : pbd_set_level_features(dev_->h, (int)l, features[l].ptr<float>()));
      dev_->fp_feat[l] = fp;
    }
    dev_->check(pbd_pdf(dev_->h));
    responses.assign(features.size(), vectorMat(nfilters_));
    dev_->fp_resp.assign(features.size(), std::vector<uint64_t>(nfilters_, 0));
    dev_->tables_resident = false;
    for (size_t l = 0; l < features.size(); ++l)
      for (size_t n = 0; n < nfilters_; ++n) {
        responses[l][n].create(features[l].rows, features[l].cols / 32, f64 ? PBD_64F : PBD_32F);
        if (!responses[l][n].empty())
          dev_->check(f64 ? pbd_get_level_response_f64(dev_->h, (int)l, (int)n, responses[l][n].ptr<double>())
                          : pbd_get_level_response(dev_->h, (int)l, (int)n, responses[l][n].ptr<float>()));
        dev_->fp_resp[l][n] = fingerprint(responses[l][n].ptr<uint8_t>(), responses[l][n].bytes());
      }
  }
};

// ---- include/DynamicProgram.hpp:61-77 ----------------------------------------------------------
static inline void append_candidates(std::vector<Candidate>& out, const std::vector<pbd_candidate_head>& heads,
                                     const std::vector<int32_t>& boxes, const std::vector<int32_t>& locs, int n, int mp) {
  for (int i = 0; i < n; ++i) {                    // src/DynamicProgram.cpp:216-251 (appends)
    Candidate c;
    c.setComponent(heads[i].component);
    c.level = heads[i].level;
    for (int p = 0; p < heads[i].nparts; ++p) {
      const int32_t* b = &boxes[((size_t)i * mp + p) * 4];
      c.addPart(Rect{b[0], b[1], b[2], b[3]}, p == 0 ? heads[i].score : 0.0f);
      for (int k = 0; k < 3; ++k) c.locs.push_back(locs[((size_t)i * mp + p) * 3 + k]);
    }
    out.push_back(c);
  }
}

// ---- include/Parts.hpp:51-261 (index tables only: all numerics run on the device) -------------------------
class Parts {
  vector3Di filterid_; vector2Di parentid_;
 public:
  Parts() {}
  explicit Parts(Model& m) : filterid_(m.filterid()), parentid_(m.parentid()) {}
  int ncomponents() const { return (int)filterid_.size(); }                  // :236
  int nparts(int c) const { return (int)filterid_[c].size(); }               // :238
  int nmixtures(int c, int p) const { return (int)filterid_[c][p].size(); }  // ComponentPart::nmixtures, :131
  int parent(int c, int p) const { return p ? parentid_[c][p] : -1; }        // ComponentPart::parent, :143
};

// DynamicProgram<T> with the reference's signatures (include/DynamicProgram.hpp:74-75).  min() transforms the SCORES
// IT IS GIVEN (planes the device does not already hold are uploaded), leaves the tables on the device and materialises
// them in the reference's shapes — rootv / rooti[level][component] and, unless fetchPointerTables(false), Ix / Iy /
// Ik[level][component][part][parent mixture] as CV_32S-like maps (src/DynamicProgram.cpp:72-76,147-151; 250 MB for the
// person model at 640x480, which is why a caller that only goes on to argmin() may switch them off).  argmin()
// back-tracks on the device from THE TABLES IT IS GIVEN: tables whose bytes are the ones min() handed out are
// already there, anything else (another engine's min(), tables edited in between) is uploaded first
// (pbd_set_root / pbd_set_dp_pointers).  Empty pointer tables (fetchPointerTables(false)) stand for "the tables of
// this object's last min()"; if there was none, argmin() throws instead of answering from whatever is resident.
template <typename T>
class DynamicProgram {
  std::shared_ptr<Device> dev_;
  bool fetch_ptr_ = true;
  static int set_resp(pbd_handle* h, int l, int n, const float* p) { return pbd_set_level_response(h, l, n, p); }
  static int set_resp(pbd_handle* h, int l, int n, const double* p) { return pbd_set_level_response_f64(h, l, n, p); }
  static int set_root(pbd_handle* h, int l, int c, const float* v, const int32_t* i) { return pbd_set_root(h, l, c, v, i); }
  static int set_root(pbd_handle* h, int l, int c, const double* v, const int32_t* i) { return pbd_set_root_f64(h, l, c, v, i); }
  static uint64_t fp3(const Mat& a, const Mat& b, const Mat& c) {
    return fingerprint(a.ptr<uint8_t>(), a.bytes()) ^ (fingerprint(b.ptr<uint8_t>(), b.bytes()) * 3) ^ (fingerprint(c.ptr<uint8_t>(), c.bytes()) * 5);
  }
 public:
  DynamicProgram() {}
  explicit DynamicProgram(std::shared_ptr<Device> d) : dev_(d) {}
  void fetchPointerTables(bool on) { fetch_ptr_ = on; }
  void min(Parts& parts, vector2DMat& scores, vector4DMat& Ix, vector4DMat& Iy, vector4DMat& Ik, vector2DMat& rootv,
           vector2DMat& rooti) {
    const size_t nscales = scores.size();
    const int ncomponents = parts.ncomponents();
    if (nscales != dev_->cell_w.size())
      throw Exception(PBD_ERR_STATE, "min(scores): level count differs from the frame geometry of the device (run HipHOGFeatures::pyramid or Device::beginFrame first)");
    if (dev_->fp_resp.size() != nscales) dev_->fp_resp.assign(nscales, std::vector<uint64_t>());
    for (size_t n = 0; n < nscales; ++n) {                        // the scores that are transformed are the ones passed in
      dev_->fp_resp[n].resize(scores[n].size(), 0);
      for (size_t f = 0; f < scores[n].size(); ++f) {
        const Mat& sc = scores[n][f];
        if (sc.rows != dev_->cell_h[n] || sc.cols != dev_->cell_w[n]) throw Exception(PBD_ERR_ARG, "min(scores): a score map has the wrong size for the frame geometry of the device");
        if (sc.empty()) continue;
        if (sc.depth() != (int)DataType<T>::type) throw Exception(PBD_ERR_ARG, "min(scores): element type differs from the detector's T");
        const uint64_t fp = fingerprint(sc.ptr<uint8_t>(), sc.bytes());
        if (fp == dev_->fp_resp[n][f]) continue;
        dev_->check(set_resp(dev_->h, (int)n, (int)f, sc.template ptr<T>()));
        dev_->fp_resp[n][f] = fp;
      }
    }
    dev_->check(pbd_dp_min(dev_->h));
    dev_->dropStale();   // compact memory plan: min() reused the feature memory and transformed the responses in place
    Ix.assign(nscales, vector3DMat(ncomponents)); Iy.assign(nscales, vector3DMat(ncomponents)); Ik.assign(nscales, vector3DMat(ncomponents));
    rootv.assign(nscales, vectorMat(ncomponents));
    rooti.assign(nscales, vectorMat(ncomponents));
    dev_->fp_root.assign(nscales, std::vector<uint64_t>(ncomponents, 0));
    dev_->fp_tab.clear();
    dev_->tables_resident = true;
    for (size_t n = 0; n < nscales; ++n)
      for (int c = 0; c < ncomponents; ++c) {
        const int rows = dev_->cell_h[n], cols = dev_->cell_w[n];
        rootv[n][c].create(rows, cols, DataType<T>::type);
        rooti[n][c].create(rows, cols, PBD_32S);
        if (!rootv[n][c].empty()) dev_->check(get_root(dev_->h, (int)n, c, rootv[n][c].template ptr<T>(), rooti[n][c].ptr<int32_t>()));
        dev_->fp_root[n][c] = fingerprint(rootv[n][c].template ptr<uint8_t>(), rootv[n][c].bytes()) ^ (fingerprint(rooti[n][c].ptr<uint8_t>(), rooti[n][c].bytes()) * 3);
        Ix[n][c].resize(parts.nparts(c)); Iy[n][c].resize(parts.nparts(c)); Ik[n][c].resize(parts.nparts(c));   // :89-91
        for (int p = 1; p < parts.nparts(c) && fetch_ptr_; ++p) {
          const int L = parts.nmixtures(c, parts.parent(c, p));
          Ix[n][c][p].resize(L); Iy[n][c][p].resize(L); Ik[n][c][p].resize(L);
          for (int m = 0; m < L; ++m) {
            Ix[n][c][p][m].create(rows, cols, PBD_32S); Iy[n][c][p][m].create(rows, cols, PBD_32S); Ik[n][c][p][m].create(rows, cols, PBD_32S);
            if (rows > 0 && cols > 0)
              dev_->check(pbd_get_dp_pointers(dev_->h, (int)n, c, p, m, Ix[n][c][p][m].ptr<int32_t>(), Iy[n][c][p][m].ptr<int32_t>(),
                                              Ik[n][c][p][m].ptr<int32_t>()));
            dev_->fp_tab.push_back(fp3(Ix[n][c][p][m], Iy[n][c][p][m], Ik[n][c][p][m]));
          }
        }
      }
  }
  static int get_root(pbd_handle* h, int l, int c, float* v, int32_t* i) { return pbd_get_root(h, l, c, v, i); }
  static int get_root(pbd_handle* h, int l, int c, double* v, int32_t* i) { return pbd_get_root_f64(h, l, c, v, i); }
  void argmin(Parts& parts, const vector2DMat& rootv, const vector2DMat& rooti, const vectorf scales,
              const vector4DMat& Ix, const vector4DMat& Iy, const vector4DMat& Ik, vectorCandidate& candidates) {
    const size_t nscales = dev_->cell_w.size();
    if (rootv.size() != nscales || rooti.size() != nscales || Ix.size() != nscales || Iy.size() != nscales || Ik.size() != nscales)
      throw Exception(PBD_ERR_ARG, "argmin(): table level count differs from the frame geometry of the device");
    // boxes are scaled by scales[n] (src/DynamicProgram.cpp:198,238): the device scales them by its pyramid's
    if (scales.size() != nscales || memcmp(scales.data(), dev_->scales.data(), nscales * sizeof(float)))
      throw Exception(PBD_ERR_UNSUPPORTED, "argmin(): `scales` differ from the scales of the device's pyramid geometry");
    if (dev_->fp_root.size() != nscales) dev_->fp_root.assign(nscales, std::vector<uint64_t>(parts.ncomponents(), 0));
    size_t t = 0;
    for (size_t n = 0; n < nscales; ++n)
      for (int c = 0; c < parts.ncomponents(); ++c) {
        const Mat &rv = rootv[n][c], &ri = rooti[n][c];
        if (rv.rows != dev_->cell_h[n] || rv.cols != dev_->cell_w[n] || ri.rows != rv.rows || ri.cols != rv.cols)
          throw Exception(PBD_ERR_ARG, "argmin(): a root table has the wrong size for the frame geometry of the device");
        if (!rv.empty()) {
          if (rv.depth() != (int)DataType<T>::type || ri.depth() != PBD_32S) throw Exception(PBD_ERR_ARG, "argmin(): root table element types");
          const uint64_t fp = fingerprint(rv.template ptr<uint8_t>(), rv.bytes()) ^ (fingerprint(ri.ptr<uint8_t>(), ri.bytes()) * 3);
          if (fp != dev_->fp_root[n][c]) {
            dev_->check(set_root(dev_->h, (int)n, c, rv.template ptr<T>(), ri.ptr<int32_t>()));
            dev_->fp_root[n][c] = fp;
          }
        }
        for (int p = 1; p < parts.nparts(c); ++p) {
          const int L = parts.nmixtures(c, parts.parent(c, p));
          const bool given = (int)Ix[n][c].size() > p && (int)Ix[n][c][p].size() == L && (int)Iy[n][c][p].size() == L && (int)Ik[n][c][p].size() == L;
          if (!given) {   // no table passed for this part: only legitimate for the tables this object's min() left on the device
            if (!dev_->tables_resident) throw Exception(PBD_ERR_STATE, "argmin(): empty pointer tables and no min() of this engine to take them from");
            continue;
          }
          for (int m = 0; m < L; ++m, ++t) {
            const Mat &x = Ix[n][c][p][m], &y = Iy[n][c][p][m], &k = Ik[n][c][p][m];
            if (x.rows != dev_->cell_h[n] || x.cols != dev_->cell_w[n] || y.rows != x.rows || y.cols != x.cols || k.rows != x.rows || k.cols != x.cols)
              throw Exception(PBD_ERR_ARG, "argmin(): a pointer table has the wrong size for the frame geometry of the device");
            if (x.empty()) continue;
            const uint64_t fp = fp3(x, y, k);
            if (t < dev_->fp_tab.size() && dev_->fp_tab[t] == fp) continue;
            dev_->check(pbd_set_dp_pointers(dev_->h, (int)n, c, p, m, x.ptr<int32_t>(), y.ptr<int32_t>(), k.ptr<int32_t>()));
            if (dev_->fp_tab.size() <= t) dev_->fp_tab.resize(t + 1, 0);
            dev_->fp_tab[t] = fp;
          }
        }
      }
    const int capacity = dev_->max_candidates, mp = pbd_max_parts(dev_->h);
    std::vector<pbd_candidate_head> heads(capacity);
    std::vector<int32_t> boxes((size_t)capacity * mp * 4), locs((size_t)capacity * mp * 3);
    int n = 0;
    dev_->check(pbd_dp_argmin(dev_->h, heads.data(), boxes.data(), locs.data(), capacity, &n));
    append_candidates(candidates, heads, boxes, locs, n, mp);      // appends, like :246-251
  }
};

// ---- include/PartsBasedDetector.hpp:152-175 ----------------------------------------------------
template <typename T>
class PartsBasedDetector {
  std::string name_;
  std::shared_ptr<Device> dev_;
  std::unique_ptr<IFeatures> features_;
  std::unique_ptr<IConvolutionEngine> convolution_engine_;
  DynamicProgram<T> dp_;
  Parts parts_;
  int device_, conv_mode_, ncomponents_ = 0;
 public:
  int max_candidates_ = 4096;
  explicit PartsBasedDetector(int device = 0, int conv_mode = PBD_CONV_AUTO, int max_candidates = 4096)
      : device_(device), conv_mode_(conv_mode), max_candidates_(max_candidates) {}
  Device& device() { return *dev_; }
  const std::string& name() const { return name_; }
  IFeatures& features() { return *features_; }
  IConvolutionEngine& convolutionEngine() { return *convolution_engine_; }
  DynamicProgram<T>& dp() { return dp_; }
  Parts& parts() { return parts_; }
  int ncomponents() const { return ncomponents_; }
  void distributeModel(Model& model) {             // src/PartsBasedDetector.cpp:102-127
    name_ = model.name();
    ncomponents_ = model.ncomponents();
    // DataType<T>::type selects the instantiation (:110,113-117)
    dev_ = std::make_shared<Device>(model, device_, conv_mode_, (int)DataType<T>::scalar, max_candidates_);
    features_.reset(new HipHOGFeatures(dev_, model.binsize(), model.nscales()));
    convolution_engine_.reset(new HipConvolutionEngine(dev_));
    convolution_engine_->setFilters(model.filters());
    parts_ = Parts(model);                         // :121-122
    dp_ = DynamicProgram<T>(dev_);
  }
  void detect(const Mat& im, vectorCandidate& candidates) { detect(im, Mat(), candidates); }
  // src/PartsBasedDetector.cpp:69-95: fused path, everything stays in HBM; `depth` ignored (:91-93)
  void detect(const Mat& im, const Mat& /*depth*/, vectorCandidate& candidates) {
    if (!dev_) throw Exception(PBD_ERR_STATE, "detect() before distributeModel()");
    const int cap = dev_->max_candidates, mp = pbd_max_parts(dev_->h);
    std::vector<pbd_candidate_head> heads(cap);
    std::vector<int32_t> boxes((size_t)cap * mp * 4), locs((size_t)cap * mp * 3);
    int n = 0;
    // (CV_8U forwards to pbd_detect_u8; CV_16U / CV_32F / CV_64F: src/HOGFeatures.cpp:136-146; anything else: PBD_ERR_UNSUPPORTED = StsUnsupportedFormat)
    dev_->check(pbd_detect_image(dev_->h, im.ptr<uint8_t>(), im.depth(), im.cols, im.rows, im.channels(), (int)im.step(),
                                 heads.data(), boxes.data(), locs.data(), cap, &n));
    append_candidates(candidates, heads, boxes, locs, n, mp);
  }
};

inline void Candidate::sort(std::vector<Candidate>& c) {
  std::stable_sort(c.begin(), c.end(), [](const Candidate& a, const Candidate& b) { return a.score() > b.score(); });
}
inline void Candidate::nonMaximaSuppression(int im_w, int im_h, std::vector<Candidate>& c, float overlap) {
  std::vector<uint8_t> scratch((size_t)im_w * im_h, 0);
  size_t keep = 0;
  for (size_t n = 0; n < c.size(); ++n) {
    Rect b = c[n].boundingBox();
    int x1 = std::max(b.x, 0), y1 = std::max(b.y, 0);
    int w = std::min(b.x + b.width, im_w) - x1, h = std::min(b.y + b.height, im_h) - y1;
    if (w <= 0 || h <= 0) x1 = y1 = w = h = 0;
    double sum = 0;
    for (int y = y1; y < y1 + h; ++y) for (int x = x1; x < x1 + w; ++x) sum += scratch[(size_t)y * im_w + x];
    if (sum / (double)(w * h) > (double)overlap) continue;
    for (int y = y1; y < y1 + h; ++y) memset(&scratch[(size_t)y * im_w + x1], 1, w);
    if (keep != n) c[keep] = c[n];
    keep++;
  }
  c.resize(keep);
}

}  // namespace pbd
#endif  // PBD_HOST_HPP_
