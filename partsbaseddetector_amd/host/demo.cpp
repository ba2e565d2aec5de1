// demo.cpp — the reference's canonical caller (src/demo.cpp:55-118) against the MI355X path:
//   pbd_demo model.bin image.raw width height channels [stagewise|double|stagewise-double]
//   pbd_demo model.bin image.raw width height channels perturb-features <responses-out.bin>
//   pbd_demo model.bin image.raw width height channels oracle-responses <responses-in.bin>
// deserialize -> distributeModel -> detect -> Candidate::sort, then prints the candidates (the
// reference shows them in a window; here they go to stdout so tests can compare them).
// `stagewise` walks pyramid -> pdf -> min -> argmin through the interface classes instead of the
// fused detect() (src/PartsBasedDetector.cpp:73-89); `double` runs PartsBasedDetector<double> like the
// ROS node and the ecto cell (ros/Node.hpp:121, cells/detect.cpp:93) instead of <float> (src/demo.cpp:85).
#include <cstdio>
#include <cstdlib>
#include <memory>
#include "pbd_filestorage.hpp"
using namespace pbd;

// The stage interfaces process THEIR ARGUMENTS (include/IConvolutionEngine.hpp:56, include/DynamicProgram.hpp:74-75):
//   perturb:  the feature pyramid is halved in place between pyramid() and pdf(); the responses of the first and the
//             last level go to `io_file` (the test compares them with the oracle's filter bank on the halved features);
//   oracle:   the scores min() transforms are read from `io_file` (random planes written by the test) instead of coming
//             from pdf(), and the tables min() returns are edited before argmin() (a root score raised, part 1's x
//             pointer at that cell redirected): the candidates must be those of the edited tables.
template <typename T>
static void run(Model& model, const Mat& im, bool stagewise, int special = 0, const char* io_file = nullptr) {
  PartsBasedDetector<T> pbd(0, PBD_CONV_EXACT);
  pbd.distributeModel(model);
  vectorCandidate candidates;
  if (stagewise) {
    vectorMat pyramid;
    pbd.features().pyramid(im, pyramid);
    vector2DMat pdf, rootv, rooti;
    vector4DMat Ix, Iy, Ik;
    if (special == 1)
      for (Mat& f : pyramid)
        for (int i = 0; i < f.rows * f.cols; ++i) f.ptr<T>()[i] *= (T)0.5;
    if (special == 2) {
      FILE* f = fopen(io_file, "rb");
      if (!f) { fprintf(stderr, "cannot open %s\n", io_file); exit(6); }
      pdf.assign(pyramid.size(), vectorMat(model.filters().size()));
      for (size_t l = 0; l < pyramid.size(); ++l)
        for (Mat& r : pdf[l]) {
          r.create(pyramid[l].rows, pyramid[l].cols / 32, DataType<T>::type);
          if (!r.empty() && fread(r.ptr<T>(), sizeof(T), (size_t)r.rows * r.cols, f) != (size_t)r.rows * r.cols) { fprintf(stderr, "short read\n"); exit(6); }
        }
      fclose(f);
    } else {
      pbd.convolutionEngine().pdf(pyramid, pdf);                                // src/PartsBasedDetector.cpp:78
    }
    if (special == 1) {
      FILE* f = fopen(io_file, "wb");
      for (size_t l : {(size_t)0, pyramid.size() - 1})
        for (const Mat& r : pdf[l]) fwrite(r.ptr<T>(), sizeof(T), (size_t)r.rows * r.cols, f);
      fclose(f);
    }
    pbd.dp().min(pbd.parts(), pdf, Ix, Iy, Ik, rootv, rooti);                   // :83, the reference's signature
    if (special == 2) {
      rootv[0][0].template at<T>(0, 0) = (T)1e6;
      for (Mat& x : Ix[0][0][1]) x.at<int32_t>(0, 0) = x.cols - 1;
    }
    // the tables came back in the reference's shapes: [level][component][part][parent mixture]
    if (Ix.size() != pyramid.size() || Ix[0][0].size() != (size_t)pbd.parts().nparts(0) || !Ix[0][0][0].empty() ||
        Ix[0][0][1].empty() || Ix[0][0][1][0].rows != pdf[0][0].rows || rootv[0][0].cols != pdf[0][0].cols) {
      fprintf(stderr, "min(): unexpected table shapes\n");
      exit(5);
    }
    // position-weighted checksum of every pointer table, in [level][component][part][parent mixture] order (Ix, Iy, Ik
    // interleaved per table): the test compares it with the oracle's tables
    unsigned long long sum = 0, idx = 0;
    for (size_t n = 0; n < Ix.size(); ++n)
      for (size_t c = 0; c < Ix[n].size(); ++c)
        for (size_t p = 1; p < Ix[n][c].size(); ++p)
          for (size_t m = 0; m < Ix[n][c][p].size(); ++m)
            for (const Mat* t : {&Ix[n][c][p][m], &Iy[n][c][p][m], &Ik[n][c][p][m]})
              for (int i = 0; i < t->rows * t->cols; ++i) sum += (++idx) * (unsigned long long)(unsigned)t->ptr<int32_t>()[i];
    printf("Tables: %llu\n", sum);
    pbd.dp().argmin(pbd.parts(), rootv, rooti, pbd.features().scales(), Ix, Iy, Ik, candidates);   // :89
  } else {
    Mat depth;
    pbd.detect(im, depth, candidates);
  }
  printf("Number of candidates: %ld\n", (long)candidates.size());
  Candidate::sort(candidates);
  for (const Candidate& c : candidates) {
    printf("%.9g %d %d", c.score(), c.component(), c.level);
    for (const Rect& r : c.parts()) printf(" %d,%d,%d,%d", r.x, r.y, r.width, r.height);
    printf("\n");
  }
}

int main(int argc, char** argv) {
  if (argc < 6 || argc > 8) {
    printf("Usage: pbd_demo model_file image.raw width height channels [stagewise|double|stagewise-double]\n");
    exit(-1);
  }
  // determine the type of model to read (src/demo.cpp:63-82)
  std::unique_ptr<Model> modelp;
  const std::string mf = argv[1];
  const std::string ext = mf.find('.') == std::string::npos ? "" : mf.substr(mf.rfind('.'));
  if (ext == ".xml" || ext == ".yaml" || ext == ".yml") modelp.reset(new FileStorageModel);
  else if (ext == ".bin") modelp.reset(new BinaryModel);
  else { printf("Unsupported model format: %s\n", ext.c_str()); exit(-2); }
  if (!modelp->deserialize(argv[1])) { printf("Error deserializing file\n"); exit(-3); }
  Model& model = *modelp;
  const int w = atoi(argv[3]), h = atoi(argv[4]), cn = atoi(argv[5]);
  Mat im(h, w, PBD_8U, cn);
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(im.ptr<uint8_t>(), 1, (size_t)w * h * cn, f) != (size_t)w * h * cn) {
    printf("Image not found or invalid image format\n");
    exit(-4);
  }
  fclose(f);
  const std::string mode = argc >= 7 ? argv[6] : "";
  const int special = mode == "perturb-features" ? 1 : mode == "oracle-responses" ? 2 : 0;
  if (special && argc != 8) { printf("%s needs a file argument\n", mode.c_str()); exit(-1); }
  const bool stagewise = special || mode.find("stagewise") != std::string::npos;
  try {
    if (mode.find("double") != std::string::npos) run<double>(model, im, stagewise);
    else run<float>(model, im, stagewise, special, special ? argv[7] : nullptr);
  } catch (const Exception& e) {
    printf("error %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
