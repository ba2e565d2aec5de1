// modelconv.cpp — model format converter, the counterpart of the reference's ModelTransfer CLI
// (src/ModelTransfer.cpp:44-74):  pbd_modelconv [--literal-defid] <in> <out>   with formats chosen by extension:
// .xml / .yaml / .yml = cv::FileStorage layout (pbd::FileStorageModel), .bin = flat dump (pbd::BinaryModel).
#include <cstdio>
#include <memory>
#include "pbd_filestorage.hpp"
using namespace pbd;

static bool is_fs(const std::string& f) {
  auto ends = [&](const char* e) { std::string s(e); return f.size() >= s.size() && f.compare(f.size() - s.size(), s.size(), s) == 0; };
  return ends(".xml") || ends(".yaml") || ends(".yml");
}
int main(int argc, char** argv) {
  bool literal = false;          // --literal-defid: read `defid` as src/FileStorageModel.cpp:148-152 does (scalar int kept, a sequence becomes {0})
  if (argc > 1 && std::string(argv[1]) == "--literal-defid") { literal = true; --argc; ++argv; }
  if (argc != 3) { printf("Usage: pbd_modelconv [--literal-defid] model_in model_out   (.xml|.yaml|.yml|.bin)\n"); return -1; }
  std::unique_ptr<Model> in;
  if (is_fs(argv[1])) { FileStorageModel* fsm = new FileStorageModel; fsm->setLiteralDefid(literal); in.reset(fsm); } else in.reset(new BinaryModel);
  if (!in->deserialize(argv[1])) { printf("Error deserializing file\n"); return -3; }
  bool ok;
  if (is_fs(argv[2])) { FileStorageModel out; out.assign(*in); ok = out.serialize(argv[2]); }
  else { BinaryModel out; out.assign(*in); ok = out.serialize(argv[2]); }
  if (!ok) { printf("Error serializing file\n"); return -4; }
  printf("%d filters, %d components, interval %d, sbin %d, thresh %g\n", (int)in->filters().size(), in->ncomponents(),
         in->nscales(), in->binsize(), in->thresh());
  return 0;
}
