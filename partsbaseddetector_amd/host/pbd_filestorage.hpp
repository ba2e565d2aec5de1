// pbd_filestorage.hpp — reader/writer for the reference's on-disk model format without OpenCV.
//
// The reference stores models with cv::FileStorage (XML or YAML); the node layout is fixed by
// FileStorageModel::serialize / deserialize (src/FileStorageModel.cpp:42-159):
//   name, interval, thresh, sbin, norient, flen            scalars
//   filtersw   sequence of opencv-matrix (rows, cols, dt, data), kh x (kw*flen), interleaved
//   biasw      sequence of reals
//   anchors    sequence of [x y] points
//   defs       sequence of [wxx wx wyy wy]
//   indexers / component-<c> / part-<p> / { parentid (int), filterid, biasid, defid (int sequences) }
// pbd::FileStorageModel mirrors the reference class (include/FileStorageModel.hpp:46-55): same
// deserialize()/serialize() signatures and return convention (bool).  The text formats follow the
// OpenCV 2.4 persistence writer (XML: <opencv_storage>, sequences as <_> elements or
// whitespace-separated scalars, matrices with type_id="opencv-matrix"; YAML: "%YAML:1.0", block maps
// by indentation, "- " block sequences, [ ... ] flow sequences, "!!opencv-matrix").
// PARITY UNPINNED: no model file ships with the reference (models/ is an un-vendored submodule,
// .gitmodules:1-3) and OpenCV is not available here, so the reader is pinned only against this
// writer and against hand-written samples in OpenCV's layout (tests/test_model_io.py).
//
// One deliberate difference BY DEFAULT: deserialize (:148-152) keeps `defid` only when the node is a scalar
// int and otherwise stores {0} — which silently breaks every part with more than one mixture
// (defid is then a K-element sequence).  Here sequences are read as sequences (the evident
// intent); a scalar still yields a 1-element vector and an empty node yields {0} like the reference.
// setLiteralDefid(true) (pbd_modelconv --literal-defid) reads the field exactly as the reference does.
#ifndef PBD_FILESTORAGE_HPP_
#define PBD_FILESTORAGE_HPP_

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include "pbd_host.hpp"

namespace pbd {

// ---- a tiny persistence tree ---------------------------------------------------------------------
struct FsNode {
  enum Kind { NONE, SCALAR, SEQ, MAP } kind = NONE;
  std::string text;                                   // SCALAR
  std::vector<FsNode> seq;                            // SEQ
  std::vector<std::pair<std::string, FsNode>> map;    // MAP (ordered)
  bool is_matrix = false;
  const FsNode* get(const std::string& k) const {
    for (auto& kv : map) if (kv.first == k) return &kv.second;
    return nullptr;
  }
  bool isInt() const {                                // cv::FileNode::isInt()
    if (kind != SCALAR || text.empty()) return false;
    char* e = nullptr; std::strtol(text.c_str(), &e, 10);
    return e && *e == 0;
  }
  double real() const { return kind == SCALAR ? std::atof(text.c_str()) : 0.0; }
  // cv::FileNode >> std::vector<T>: a scalar reads as a 1-element sequence
  template <typename T> std::vector<T> numbers() const {
    std::vector<T> out;
    if (kind == SCALAR) { if (!text.empty()) out.push_back((T)std::atof(text.c_str())); }
    else if (kind == SEQ) for (auto& n : seq) { auto v = n.numbers<T>(); out.insert(out.end(), v.begin(), v.end()); }
    return out;
  }
};

static inline std::vector<std::string> fs_split_ws(const std::string& s) {
  std::vector<std::string> t; std::istringstream is(s); std::string w;
  while (is >> w) t.push_back(w);
  return t;
}
static inline FsNode fs_scalars_to_node(const std::vector<std::string>& toks) {
  FsNode n;
  if (toks.empty()) { n.kind = FsNode::SEQ; return n; }          // empty sequence
  if (toks.size() == 1) { n.kind = FsNode::SCALAR; n.text = toks[0]; return n; }
  n.kind = FsNode::SEQ;
  for (auto& t : toks) { FsNode c; c.kind = FsNode::SCALAR; c.text = t; n.seq.push_back(c); }
  return n;
}

// ---- XML -------------------------------------------------------------------------------------------
class FsXml {
  const std::string& s; size_t p = 0;
  void ws() { while (p < s.size() && std::isspace((unsigned char)s[p])) ++p; }
  bool starts(const char* t) const { return s.compare(p, strlen(t), t) == 0; }
  void skip_misc() {
    for (;;) {
      ws();
      if (starts("<?")) { p = s.find("?>", p); p = (p == std::string::npos) ? s.size() : p + 2; }
      else if (starts("<!--")) { p = s.find("-->", p); p = (p == std::string::npos) ? s.size() : p + 3; }
      else break;
    }
  }
  static std::string unescape(const std::string& t) {
    std::string o;
    for (size_t i = 0; i < t.size(); ++i) {
      if (t[i] == '&') {
        if (!t.compare(i, 4, "&lt;")) { o += '<'; i += 3; } else if (!t.compare(i, 4, "&gt;")) { o += '>'; i += 3; }
        else if (!t.compare(i, 5, "&amp;")) { o += '&'; i += 4; } else if (!t.compare(i, 6, "&quot;")) { o += '"'; i += 5; }
        else o += t[i];
      } else o += t[i];
    }
    return o;
  }
  // parses one element starting at '<'; returns its name and node
  bool element(std::string& name, FsNode& node) {
    if (p >= s.size() || s[p] != '<') return false;
    size_t e = s.find('>', p);
    if (e == std::string::npos) return false;
    std::string head = s.substr(p + 1, e - p - 1);
    bool selfclose = !head.empty() && head.back() == '/';
    if (selfclose) head.pop_back();
    std::istringstream hs(head); hs >> name;
    const bool matrix = head.find("opencv-matrix") != std::string::npos;
    p = e + 1;
    if (selfclose) { node.kind = FsNode::SEQ; return true; }
    std::string text; std::vector<std::pair<std::string, FsNode>> kids;
    for (;;) {
      skip_misc();
      if (p >= s.size()) return false;
      if (starts("</")) { p = s.find('>', p); if (p == std::string::npos) return false; ++p; break; }
      if (s[p] == '<') { std::string cn; FsNode c; if (!element(cn, c)) return false; kids.push_back({cn, c}); }
      else { size_t q = s.find('<', p); text += s.substr(p, q - p) + " "; p = q; }
    }
    if (kids.empty()) {
      std::string u = unescape(text);
      auto toks = fs_split_ws(u);
      if (!u.empty() && u.find('"') != std::string::npos) {        // quoted string scalar
        size_t a = u.find('"'), b = u.rfind('"');
        node.kind = FsNode::SCALAR; node.text = u.substr(a + 1, b - a - 1);
      } else node = fs_scalars_to_node(toks);
    } else {
      bool all_anon = true;
      for (auto& k : kids) if (k.first != "_") all_anon = false;
      if (all_anon) { node.kind = FsNode::SEQ; for (auto& k : kids) node.seq.push_back(k.second); }
      else { node.kind = FsNode::MAP; node.map = kids; }
    }
    node.is_matrix = matrix;
    return true;
  }
 public:
  explicit FsXml(const std::string& src) : s(src) {}
  bool parse(FsNode& root) {
    skip_misc();
    std::string name;
    return element(name, root) && name == "opencv_storage";
  }
};

// ---- YAML (the subset the OpenCV writer emits) ---------------------------------------------------------
class FsYaml {
  struct Line { int indent; std::string text; };
  std::vector<Line> L; size_t i = 0;
  static std::string trim(const std::string& t) {
    size_t a = t.find_first_not_of(" \t\r"), b = t.find_last_not_of(" \t\r");
    return a == std::string::npos ? "" : t.substr(a, b - a + 1);
  }
  static FsNode flow(const std::string& t0) {           // "[ a, b, [c, d] ]" or a scalar
    std::string t = trim(t0);
    if (t.size() >= 2 && t[0] == '"' && t.back() == '"') { FsNode n; n.kind = FsNode::SCALAR; n.text = t.substr(1, t.size() - 2); return n; }
    if (t.empty() || t[0] != '[') { FsNode n; n.kind = t.empty() ? FsNode::NONE : FsNode::SCALAR; n.text = t; return n; }
    FsNode n; n.kind = FsNode::SEQ;
    int depth = 0; std::string cur;
    for (size_t k = 1; k + 1 < t.size() || (k < t.size() && t[k] != ']'); ++k) {
      char c = t[k];
      if (c == '[') depth++;
      if (c == ']') { if (depth == 0) break; depth--; }
      if (c == ',' && depth == 0) { if (!trim(cur).empty()) n.seq.push_back(flow(cur)); cur.clear(); }
      else cur += c;
    }
    if (!trim(cur).empty()) n.seq.push_back(flow(cur));
    return n;
  }
  // a value that may continue over following, deeper-indented lines (long flow sequences)
  std::string gather(std::string first, int indent) {
    int open = 0;
    for (char c : first) { if (c == '[') open++; if (c == ']') open--; }
    while (open > 0 && i < L.size()) {
      for (char c : L[i].text) { if (c == '[') open++; if (c == ']') open--; }
      first += " " + L[i].text; ++i;
    }
    (void)indent;
    return first;
  }
  FsNode block(int indent) {                            // a block map or block sequence at this indent
    FsNode n;
    if (i >= L.size()) return n;
    if (L[i].text.compare(0, 1, "-") == 0) {
      n.kind = FsNode::SEQ;
      while (i < L.size() && L[i].indent == indent && L[i].text[0] == '-') {
        const std::string rest = trim(L[i].text.substr(1));
        ++i;
        if (rest.empty() || rest.compare(0, 2, "!!") == 0) {     // "- !!opencv-matrix" + deeper block map
          FsNode c = (i < L.size() && L[i].indent > indent) ? block(L[i].indent) : FsNode();
          c.is_matrix = !rest.empty();
          n.seq.push_back(c);
        } else {
          n.seq.push_back(flow(gather(rest, indent)));             // "- [ a, b ]" or "- scalar"
        }
      }
      return n;
    }
    n.kind = FsNode::MAP;
    while (i < L.size() && L[i].indent == indent && L[i].text[0] != '-') {
      const std::string& t = L[i].text;
      size_t c = t.find(':');
      if (c == std::string::npos) { ++i; continue; }
      std::string key = trim(t.substr(0, c)), val = trim(t.substr(c + 1));
      ++i;
      FsNode v;
      if (val.empty() || val.compare(0, 16, "!!opencv-matrix") == 0) {
        if (i < L.size() && (L[i].indent > indent || (L[i].indent == indent && L[i].text[0] == '-'))) v = block(L[i].indent);
        v.is_matrix = !val.empty();
      } else v = flow(gather(val, indent));
      n.map.push_back({key, v});
    }
    return n;
  }
 public:
  explicit FsYaml(const std::string& src) {
    std::istringstream is(src); std::string ln;
    while (std::getline(is, ln)) {
      if (ln.compare(0, 5, "%YAML") == 0 || ln.compare(0, 3, "---") == 0) continue;
      size_t a = ln.find_first_not_of(' ');
      if (a == std::string::npos || ln[a] == '#') continue;
      size_t b = ln.find_last_not_of(" \t\r");
      L.push_back(Line{(int)a, ln.substr(a, b - a + 1)});
    }
  }
  bool parse(FsNode& root) { root = block(L.empty() ? 0 : L[0].indent); return root.kind == FsNode::MAP; }
};

// ---- include/FileStorageModel.hpp:46-55 ----------------------------------------------------------------
class FileStorageModel : public Model {
  bool literal_defid_ = false;
  static bool mat_from(const FsNode& n, Mat& m) {
    const FsNode *r = n.get("rows"), *c = n.get("cols"), *d = n.get("data");
    if (!r || !c || !d) return false;
    const int rows = (int)r->real(), cols = (int)c->real();
    std::vector<double> v = d->numbers<double>();
    if ((size_t)rows * cols != v.size()) return false;
    m.create(rows, cols, PBD_32F);                       // distributeModel converts to T anyway (:114-117)
    for (size_t k = 0; k < v.size(); ++k) m.ptr<float>()[k] = (float)v[k];
    return true;
  }
 public:
  // false (default): `defid` sequences are read as sequences — the evident intent; true: src/FileStorageModel.cpp:148-152 literally
  // (`if (defid.isInt()) defid >> defid_[c][p]; else defid_[c][p].push_back(0);` — every part with more than one mixture, and in YAML
  // every part, ends up with the single deformation index 0): for a caller who wants the reference's behaviour on such a file bug for bug
  void setLiteralDefid(bool on) { literal_defid_ = on; }
  bool deserialize(const std::string& filename) override {   // src/FileStorageModel.cpp:96-159
    std::ifstream f(filename.c_str(), std::ios::binary);
    if (!f) return false;
    std::stringstream ss; ss << f.rdbuf();
    const std::string src = ss.str();
    FsNode root;
    size_t a = src.find_first_not_of(" \t\r\n");
    const bool xml = a != std::string::npos && src[a] == '<';
    if (xml) { FsXml p(src); if (!p.parse(root)) return false; }
    else { FsYaml p(src); if (!p.parse(root)) return false; }
    auto scalar = [&](const char* k, double def) { const FsNode* n = root.get(k); return n ? n->real() : def; };
    if (const FsNode* n = root.get("name")) name_ = n->text;
    nscales_ = (int)scalar("interval", 0); thresh_ = (float)scalar("thresh", 0); binsize_ = (int)scalar("sbin", 0);
    norient_ = (int)scalar("norient", 0); flen_ = (int)scalar("flen", 0);
    filtersw_.clear(); biasw_.clear(); anchors_.clear(); defw_.clear();
    const FsNode* fw = root.get("filtersw");
    if (!fw || fw->kind != FsNode::SEQ) return false;
    for (auto& n : fw->seq) { Mat m; if (!mat_from(n, m)) return false; filtersw_.push_back(m); }
    if (const FsNode* n = root.get("biasw")) biasw_ = n->numbers<float>();
    if (const FsNode* n = root.get("anchors")) {
      if (n->kind == FsNode::SEQ && !n->seq.empty() && n->seq[0].kind == FsNode::SEQ)
        for (auto& q : n->seq) { auto v = q.numbers<int>(); if (v.size() < 2) return false; anchors_.push_back(Point{v[0], v[1]}); }
      else { auto v = n->numbers<int>(); for (size_t k = 0; k + 1 < v.size(); k += 2) anchors_.push_back(Point{v[k], v[k + 1]}); }
    }
    if (const FsNode* n = root.get("defs")) {
      if (n->kind != FsNode::SEQ) return false;
      for (auto& q : n->seq) { auto v = q.numbers<float>(); if (v.size() != 4) return false; defw_.push_back(v); }
    }
    const FsNode* ix = root.get("indexers");
    if (!ix || ix->kind != FsNode::MAP) return false;
    const size_t ncomp = ix->map.size();
    parentid_.assign(ncomp, {}); filterid_.assign(ncomp, {}); biasid_.assign(ncomp, {}); defid_.assign(ncomp, {});
    for (size_t c = 0; c < ncomp; ++c) {
      std::ostringstream cs; cs << "component-" << c;
      const FsNode* comp = ix->get(cs.str());
      if (!comp || comp->kind != FsNode::MAP) return false;
      const size_t np = comp->map.size();
      parentid_[c].resize(np); filterid_[c].resize(np); biasid_[c].resize(np); defid_[c].resize(np);
      for (size_t p = 0; p < np; ++p) {
        std::ostringstream ps; ps << "part-" << p;
        const FsNode* part = comp->get(ps.str());
        if (!part) return false;
        if (const FsNode* n = part->get("parentid")) parentid_[c][p] = (int)n->real();
        if (const FsNode* n = part->get("filterid")) filterid_[c][p] = n->numbers<int>();
        if (const FsNode* n = part->get("biasid")) biasid_[c][p] = n->numbers<int>();
        const FsNode* d = part->get("defid");
        std::vector<int> dv = d ? d->numbers<int>() : std::vector<int>();
        if (dv.empty()) dv.push_back(0);                 // :151 (empty / missing node)
        if (literal_defid_)                              // :148-152 to the letter: a scalar int is kept, anything else (a K-element sequence!) becomes {0}
          dv = (d && d->isInt()) ? std::vector<int>(1, (int)d->real()) : std::vector<int>(1, 0);
        defid_[c][p] = dv;
      }
    }
    return !filtersw_.empty() && flen_ > 0;
  }

  bool serialize(const std::string& filename) const {        // src/FileStorageModel.cpp:42-94
    const bool yaml = filename.size() > 4 && (filename.rfind(".yaml") == filename.size() - 5 || filename.rfind(".yml") == filename.size() - 4);
    std::ofstream f(filename.c_str());
    if (!f) return false;
    f.precision(9);
    auto num = [](std::ostream& o, double v) { std::ostringstream t; t.precision(9); t << v; std::string s = t.str(); if (s.find_first_of(".en") == std::string::npos) s += "."; o << s; };
    if (!yaml) {
      f << "<?xml version=\"1.0\"?>\n<opencv_storage>\n<name>" << name_ << "</name>\n<interval>" << nscales_ << "</interval>\n<thresh>";
      num(f, thresh_);
      f << "</thresh>\n<sbin>" << binsize_ << "</sbin>\n<norient>" << norient_ << "</norient>\n<flen>" << flen_ << "</flen>\n<filtersw>\n";
      for (const Mat& m : filtersw_) {
        f << "  <_ type_id=\"opencv-matrix\">\n    <rows>" << m.rows << "</rows>\n    <cols>" << m.cols << "</cols>\n    <dt>f</dt>\n    <data>\n     ";
        for (int k = 0; k < m.rows * m.cols; ++k) { f << ' '; num(f, m.ptr<float>()[k]); if (k % 6 == 5) f << "\n     "; }
        f << "</data></_>\n";
      }
      f << "</filtersw>\n<biasw>\n ";
      for (float b : biasw_) { f << ' '; num(f, b); }
      f << "</biasw>\n<anchors>\n";
      for (const Point& a : anchors_) f << "  <_>\n    " << a.x << ' ' << a.y << "</_>\n";
      f << "</anchors>\n<defs>\n";
      for (const vectorf& d : defw_) { f << "  <_>\n   "; for (float v : d) { f << ' '; num(f, v); } f << "</_>\n"; }
      f << "</defs>\n<indexers>\n";
      for (size_t c = 0; c < filterid_.size(); ++c) {
        f << "  <component-" << c << ">\n";
        for (size_t p = 0; p < filterid_[c].size(); ++p) {
          f << "    <part-" << p << ">\n      <parentid>" << parentid_[c][p] << "</parentid>\n      <filterid>";
          for (int v : filterid_[c][p]) f << ' ' << v;
          f << "</filterid>\n      <biasid>";
          for (int v : biasid_[c][p]) f << ' ' << v;
          f << "</biasid>\n      <defid>";
          if (p > 0) for (int v : defid_[c][p]) f << ' ' << v;
          f << "</defid></part-" << p << ">\n";
        }
        f << "  </component-" << c << ">\n";
      }
      f << "</indexers>\n</opencv_storage>\n";
    } else {
      f << "%YAML:1.0\nname: " << name_ << "\ninterval: " << nscales_ << "\nthresh: "; num(f, thresh_);
      f << "\nsbin: " << binsize_ << "\nnorient: " << norient_ << "\nflen: " << flen_ << "\nfiltersw:\n";
      for (const Mat& m : filtersw_) {
        f << "   - !!opencv-matrix\n      rows: " << m.rows << "\n      cols: " << m.cols << "\n      dt: f\n      data: [";
        for (int k = 0; k < m.rows * m.cols; ++k) { f << ' '; num(f, m.ptr<float>()[k]); if (k + 1 < m.rows * m.cols) f << ','; if (k % 6 == 5) f << "\n         "; }
        f << " ]\n";
      }
      f << "biasw: [";
      for (size_t k = 0; k < biasw_.size(); ++k) { f << ' '; num(f, biasw_[k]); if (k + 1 < biasw_.size()) f << ','; if (k % 8 == 7) f << "\n   "; }
      f << " ]\nanchors:\n";
      for (const Point& a : anchors_) f << "   - [ " << a.x << ", " << a.y << " ]\n";
      f << "defs:\n";
      for (const vectorf& d : defw_) { f << "   - ["; for (size_t k = 0; k < d.size(); ++k) { f << ' '; num(f, d[k]); if (k + 1 < d.size()) f << ','; } f << " ]\n"; }
      f << "indexers:\n";
      for (size_t c = 0; c < filterid_.size(); ++c) {
        f << "   component-" << c << ":\n";
        for (size_t p = 0; p < filterid_[c].size(); ++p) {
          f << "      part-" << p << ":\n         parentid: " << parentid_[c][p] << "\n         filterid: [";
          for (size_t k = 0; k < filterid_[c][p].size(); ++k) f << (k ? ", " : " ") << filterid_[c][p][k];
          f << " ]\n         biasid: [";
          for (size_t k = 0; k < biasid_[c][p].size(); ++k) f << (k ? ", " : " ") << biasid_[c][p][k];
          f << " ]\n         defid: [";
          if (p > 0) for (size_t k = 0; k < defid_[c][p].size(); ++k) f << (k ? ", " : " ") << defid_[c][p][k];
          f << " ]\n";
        }
      }
    }
    return (bool)f;
  }
  // adopt the contents of another model (e.g. a BinaryModel) for serialisation
  void assign(Model& o) {
    filtersw_ = o.filters(); defw_ = o.def(); biasw_ = o.bias(); anchors_ = o.anchors(); biasid_ = o.biasid();
    filterid_ = o.filterid(); defid_ = o.defid(); parentid_ = o.parentid(); name_ = o.name(); nscales_ = o.nscales();
    thresh_ = o.thresh(); binsize_ = o.binsize(); flen_ = o.flen(); norient_ = o.norient();
  }
};

}  // namespace pbd
#endif  // PBD_FILESTORAGE_HPP_
