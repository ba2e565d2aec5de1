// demo.cpp — the reference's canonical caller (src/demo.cpp:55-118) against the MI355X path:
//   pbd_demo model.bin image.raw width height channels [stagewise|double|stagewise-double]
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

template <typename T>
static void run(Model& model, const Mat& im, bool stagewise) {
  PartsBasedDetector<T> pbd(0, PBD_CONV_EXACT);
  pbd.distributeModel(model);
  vectorCandidate candidates;
  if (stagewise) {
    vectorMat pyramid;
    pbd.features().pyramid(im, pyramid);
    vector2DMat pdf, rootv, rooti;
    vector4DMat Ix, Iy, Ik;
    pbd.convolutionEngine().pdf(pyramid, pdf);                                  // src/PartsBasedDetector.cpp:78
    pbd.dp().min(pbd.parts(), pdf, Ix, Iy, Ik, rootv, rooti);                   // :83, the reference's signature
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
  if (argc != 6 && argc != 7) {
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
  const std::string mode = argc == 7 ? argv[6] : "";
  const bool stagewise = mode.find("stagewise") != std::string::npos;
  try {
    if (mode.find("double") != std::string::npos) run<double>(model, im, stagewise);
    else run<float>(model, im, stagewise);
  } catch (const Exception& e) {
    printf("error %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
