// dump_reference.cpp — runs the REAL reference stages (wg-perception/PartsBasedDetector, linked with a real OpenCV) on
// the inputs written by make_pin_inputs.py and dumps every stage's output.  See README.md.  Never built in the graft
// image; the record format is read by pins_to_npz.py:
//   record = u32 name_len | name | u32 dtype (0 u8, 1 i32, 2 f32, 3 f64) | u32 ndim | u32 dims[ndim] | data
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Candidate.hpp"
#include "DistanceTransform.hpp"
#include "DynamicProgram.hpp"
#include "FileStorageModel.hpp"
#include "HOGFeatures.hpp"
#include "Parts.hpp"
#include "SpatialConvolutionEngine.hpp"

static FILE* g_out;
static void rec(const std::string& name, unsigned dtype, const std::vector<unsigned>& dims, const void* data, size_t bytes) {
  unsigned n = (unsigned)name.size(), nd = (unsigned)dims.size();
  fwrite(&n, 4, 1, g_out); fwrite(name.data(), 1, n, g_out); fwrite(&dtype, 4, 1, g_out); fwrite(&nd, 4, 1, g_out);
  fwrite(dims.data(), 4, nd, g_out); fwrite(data, 1, bytes, g_out);
}
static void rec_mat(const std::string& name, const cv::Mat& m0) {
  cv::Mat m = m0.isContinuous() ? m0 : m0.clone();
  const unsigned dt = m.depth() == CV_8U ? 0 : m.depth() == CV_32S ? 1 : m.depth() == CV_32F ? 2 : 3;
  rec(name, dt, {(unsigned)m.rows, (unsigned)(m.cols * m.channels())}, m.data, m.total() * m.elemSize());
}
static std::string idx(const char* base, int a, int b = -1, int c = -1, int d = -1) {
  char buf[128];
  if (d >= 0) snprintf(buf, sizeof buf, "%s_%d_%d_%d_%d", base, a, b, c, d);
  else if (c >= 0) snprintf(buf, sizeof buf, "%s_%d_%d_%d", base, a, b, c);
  else if (b >= 0) snprintf(buf, sizeof buf, "%s_%d_%d", base, a, b);
  else snprintf(buf, sizeof buf, "%s_%d", base, a);
  return buf;
}

template <typename T>
static void run_frame(const std::string& dir, const std::string& tag, Model& model, int w, int h, int cn) {
  std::vector<unsigned char> px((size_t)w * h * cn);
  std::ifstream f((dir + "/image_" + tag + ".raw").c_str(), std::ios::binary);
  f.read((char*)px.data(), px.size());
  cv::Mat im(h, w, cn == 3 ? CV_8UC3 : CV_8UC1, px.data());
  // exactly PartsBasedDetector<T>::distributeModel + detect (src/PartsBasedDetector.cpp:69-127), stage by stage
  HOGFeatures<T> features(model.binsize(), model.nscales(), model.flen(), model.norient());
  SpatialConvolutionEngine conv(cv::DataType<T>::type, model.flen());
  for (size_t n = 0; n < model.filters().size(); ++n) model.filters()[n].convertTo(model.filters()[n], cv::DataType<T>::type);
  conv.setFilters(model.filters());
  Parts parts(model.filters(), model.filtersi(), model.def(), model.defi(), model.bias(), model.biasi(), model.anchors(),
              model.biasid(), model.filterid(), model.defid(), model.parentid());
  DynamicProgram<T> dp(model.thresh());
  vectorMat pyramid;
  features.pyramid(im, pyramid);
  vectorf scales = features.scales();
  rec(tag + "_scales", 2, {(unsigned)scales.size()}, scales.data(), scales.size() * 4);
  for (size_t l = 0; l < pyramid.size(); ++l) rec_mat(idx((tag + "_feat").c_str(), (int)l), pyramid[l]);
  vector2DMat pdf;
  conv.pdf(pyramid, pdf);
  for (size_t l = 0; l < pdf.size(); ++l)
    for (size_t n = 0; n < pdf[l].size(); ++n) rec_mat(idx((tag + "_resp").c_str(), (int)l, (int)n), pdf[l][n]);
  vector4DMat Ix, Iy, Ik;
  vector2DMat rootv, rooti;
  dp.min(parts, pdf, Ix, Iy, Ik, rootv, rooti);
  for (size_t l = 0; l < rootv.size(); ++l)
    for (size_t c = 0; c < rootv[l].size(); ++c) {
      rec_mat(idx((tag + "_rootv").c_str(), (int)l, (int)c), rootv[l][c]);
      rec_mat(idx((tag + "_rooti").c_str(), (int)l, (int)c), rooti[l][c]);
      for (size_t p = 1; p < Ix[l][c].size(); ++p)
        for (size_t m = 0; m < Ix[l][c][p].size(); ++m) {
          rec_mat(idx((tag + "_Ix").c_str(), (int)l, (int)c, (int)p, (int)m), Ix[l][c][p][m]);
          rec_mat(idx((tag + "_Iy").c_str(), (int)l, (int)c, (int)p, (int)m), Iy[l][c][p][m]);
          rec_mat(idx((tag + "_Ik").c_str(), (int)l, (int)c, (int)p, (int)m), Ik[l][c][p][m]);
        }
    }
  vectorCandidate cands;
  dp.argmin(parts, rootv, rooti, scales, Ix, Iy, Ik, cands);
  std::vector<float> conf;
  std::vector<int> comp, boxes;
  for (size_t i = 0; i < cands.size(); ++i) {
    comp.push_back(cands[i].component());
    comp.push_back((int)cands[i].parts().size());
    conf.push_back(cands[i].score());
    for (size_t p = 0; p < cands[i].parts().size(); ++p) {
      const cv::Rect r = cands[i].parts()[p];
      boxes.push_back(r.x); boxes.push_back(r.y); boxes.push_back(r.width); boxes.push_back(r.height);
    }
  }
  rec(tag + "_cand_score", 2, {(unsigned)conf.size()}, conf.data(), conf.size() * 4);
  rec(tag + "_cand_comp_nparts", 1, {(unsigned)cands.size(), 2u}, comp.data(), comp.size() * 4);
  rec(tag + "_cand_boxes", 1, {(unsigned)boxes.size() / 4, 4u}, boxes.data(), boxes.size() * 4);
}

// HOGFeatures<T>::pyramid on an image of depth CV_16U / CV_32F / CV_64F (src/HOGFeatures.cpp:136-146): the feature pyramid pins the linked
// OpenCV's cv::resize / cv::pyrDown of that depth and features<IT>
template <typename T>
static void run_wide(const std::string& dir, const std::string& tag, int w, int h, int cn, int depth, int sbin, int interval) {
  const size_t esz = depth == CV_16U ? 2 : depth == CV_32F ? 4 : 8;
  std::vector<unsigned char> px((size_t)w * h * cn * esz);
  std::ifstream f((dir + "/image_" + tag + ".raw").c_str(), std::ios::binary);
  f.read((char*)px.data(), px.size());
  cv::Mat im(h, w, CV_MAKETYPE(depth, cn), px.data());
  HOGFeatures<T> features(sbin, interval, 32, 18);
  vectorMat pyramid;
  features.pyramid(im, pyramid);
  const std::string t = tag + (sizeof(T) == 4 ? "_f32" : "_f64");
  for (size_t l = 0; l < pyramid.size(); ++l) rec_mat(idx((t + "_feat").c_str(), (int)l), pyramid[l]);
}

template <typename T>
static void run_dt(const std::string& dir, int i) {
  // dt_<i>.bin: i32 rows, cols, osx, osy | f64 ax, bx, ay, by | f32 rows*cols
  std::ifstream f((dir + idx("/dt", i) + ".bin").c_str(), std::ios::binary);
  int hd[4]; double q[4];
  f.read((char*)hd, 16); f.read((char*)q, 32);
  cv::Mat_<float> in32(hd[0], hd[1]);
  f.read((char*)in32.data, (size_t)hd[0] * hd[1] * 4);
  cv::Mat_<T> in; in32.convertTo(in, cv::DataType<T>::type);
  DistanceTransform<T> dt;
  Quadratic fx(q[0], q[1]), fy(q[2], q[3]);
  cv::Mat_<T> out; cv::Mat_<int> Ix, Iy;
  dt.compute(in, fx, fy, cv::Point(hd[2], hd[3]), out, Ix, Iy);
  const std::string t = sizeof(T) == 4 ? "f32" : "f64";
  rec_mat(idx(("dt_" + t + "_out").c_str(), i), out); rec_mat(idx(("dt_" + t + "_ix").c_str(), i), Ix); rec_mat(idx(("dt_" + t + "_iy").c_str(), i), Iy);
}

int main(int argc, char** argv) {
  if (argc != 3) { printf("usage: pbd_ref_dump <inputs dir> <out.bin>\n"); return 1; }
  const std::string dir = argv[1];
  g_out = fopen(argv[2], "wb");
  if (!g_out) return 2;
  const std::string ver = PBD_OPENCV_VERSION;
  rec("opencv_version", 0, {(unsigned)ver.size()}, ver.data(), ver.size());
  // manifest.txt: lines "frame <tag> <model file> <w> <h> <cn>", "wide <tag> <w> <h> <cn> <cv depth>" and "dt <count>"
  std::ifstream mf((dir + "/manifest.txt").c_str());
  std::string kind;
  while (mf >> kind) {
    if (kind == "frame") {
      std::string tag, mfile; int w, h, cn;
      mf >> tag >> mfile >> w >> h >> cn;
      { FileStorageModel model; if (!model.deserialize(dir + "/" + mfile)) { fprintf(stderr, "cannot read %s\n", mfile.c_str()); return 3; } run_frame<float>(dir, tag + "_f32", model, w, h, cn); }
      { FileStorageModel model; model.deserialize(dir + "/" + mfile); run_frame<double>(dir, tag + "_f64", model, w, h, cn); }
    } else if (kind == "wide") {
      std::string tag; int w, h, cn, depth;
      mf >> tag >> w >> h >> cn >> depth;
      run_wide<float>(dir, tag, w, h, cn, depth, 4, 10); run_wide<double>(dir, tag, w, h, cn, depth, 4, 10);   // cell size 4, 10 levels per octave (make_tree_model's defaults: tests/test_reference_pins.py)
    } else if (kind == "dt") {
      int n; mf >> n;
      for (int i = 0; i < n; ++i) { run_dt<float>(dir, i); run_dt<double>(dir, i); }
    }
  }
  fclose(g_out);
  return 0;
}
