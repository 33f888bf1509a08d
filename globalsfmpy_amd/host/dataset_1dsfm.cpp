// 1DSfM keypoints/tracks ingestion and the CalcCovariance driver: the caller of the batched per-edge covariance kernel
// (SURVEY 8f rows 2 and 4).  Follows thirdparty/TheiaSfM/src/theia/io/read_1dsfm.cc:93-292 for the file formats and
// src/uncertainty.cpp:3-33,82-198 + bind_src/GlobalSfMpy.cpp:623-628 for what is estimated and written.
#include <glob.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../../include/gsfm/view_graph.hpp"
#include "../../include/gsfm_rot.h"

namespace gsfm {

namespace {
void AngleAxisToRowMajorMatrix(const double* a, double* R) {  // ceres::AngleAxisToRotationMatrix semantics
  const double t2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  if (t2 > 2.220446049250313e-16) {
    const double t = std::sqrt(t2), wx = a[0] / t, wy = a[1] / t, wz = a[2] / t, c = std::cos(t), s = std::sin(t), k = 1.0 - c;
    R[0] = c + wx * wx * k; R[1] = wx * wy * k - wz * s; R[2] = wy * s + wx * wz * k;
    R[3] = wz * s + wx * wy * k; R[4] = c + wy * wy * k; R[5] = -wx * s + wy * wz * k;
    R[6] = -wy * s + wx * wz * k; R[7] = wx * s + wy * wz * k; R[8] = c + wz * wz * k;
  } else {
    R[0] = 1; R[1] = -a[2]; R[2] = a[1]; R[3] = a[2]; R[4] = 1; R[5] = -a[0]; R[6] = -a[1]; R[7] = a[0]; R[8] = 1;
  }
}
}  // namespace

bool Read1DSFMTracks(const std::string& dir, Tracks1DSfM* out, std::string* error) {
  *out = Tracks1DSfM();
  std::unordered_set<theia::ViewId> cc;
  {
    std::ifstream f(dir + "/cc.txt");
    theia::ViewId v;
    while (f >> v) cc.insert(v);
  }
  {
    // list.txt: "<image path>[ 0 <focal>]" per line; the view id is the line index (read_1dsfm.cc:114-163).
    std::ifstream f(dir + "/list.txt");
    if (!f.is_open()) { if (error) *error = "cannot read " + dir + "/list.txt"; return false; }
    std::string line;
    theia::ViewId id = 0;
    while (std::getline(f, line)) {
      std::istringstream ss(line);
      std::string name;
      if (!(ss >> name)) continue;
      int flag = 0;
      double focal = 0.0;
      if (!(ss >> flag >> focal)) focal = 0.0;
      if (cc.empty() || cc.count(id)) {
        out->focal[id] = focal;
        const size_t slash = name.find_last_of('/');
        out->names[id] = slash == std::string::npos ? name : name.substr(slash + 1);
      }
      ++id;
    }
  }
  {
    // coords.txt: "#index = I, name = N, keys = K, px = X, py = Y, focal = F" then K lines "k x y 0 0 r g b" (:165-230).
    std::ifstream f(dir + "/coords.txt");
    if (!f.is_open()) { if (error) *error = "cannot read " + dir + "/coords.txt"; return false; }
    std::string line;
    while (std::getline(f, line)) {
      if (line.empty()) continue;
      unsigned id = 0;
      int keys = 0;
      float px = 0, py = 0, fl = 0;
      char name[1024];
      if (std::sscanf(line.c_str(), "#index = %u, name = %1023s keys = %d, px = %f, py = %f, focal = %f", &id, name, &keys, &px, &py, &fl) < 3) {
        if (error) *error = "malformed coords.txt header: " + line;
        return false;
      }
      const bool keep = out->focal.count(id) != 0;
      std::vector<Eigen::Vector2d>* kp = nullptr;
      if (keep) {
        out->principal_point[id] = Eigen::Vector2d(px, py);
        kp = &out->keypoints[id];
        kp->reserve(keys);
      }
      for (int k = 0; k < keys; ++k) {
        if (!std::getline(f, line)) { if (error) *error = "coords.txt ends inside a key list"; return false; }
        if (!keep) continue;
        double x = 0, y = 0;
        if (std::sscanf(line.c_str(), "%*d %lf %lf", &x, &y) != 2) { if (error) *error = "malformed coords.txt key: " + line; return false; }
        kp->emplace_back(x, y);
      }
    }
  }
  {
    // tracks.txt: "<num tracks>" then per track "<len> (<view> <key>)*len" (:232-292).
    std::ifstream f(dir + "/tracks.txt");
    if (!f.is_open()) { if (error) *error = "cannot read " + dir + "/tracks.txt"; return false; }
    size_t n = 0;
    f >> n;
    out->tracks.reserve(n);
    for (size_t t = 0; t < n; ++t) {
      int len = 0;
      if (!(f >> len)) { if (error) *error = "tracks.txt is shorter than its header says"; return false; }
      std::vector<std::pair<theia::ViewId, int>> tr;
      tr.reserve(len);
      for (int k = 0; k < len; ++k) {
        theia::ViewId v;
        int key;
        if (!(f >> v >> key)) { if (error) *error = "tracks.txt ends inside a track"; return false; }
        const auto it = out->keypoints.find(v);
        if (it == out->keypoints.end()) continue;  // view outside the component
        if (key < 0 || (size_t)key >= it->second.size()) { if (error) *error = "tracks.txt references a missing keypoint"; return false; }
        tr.emplace_back(v, key);
      }
      out->tracks.push_back(std::move(tr));
    }
  }
  return true;
}

void CollectEdgeMatches(const Tracks1DSfM& tr, const theia::ViewGraph& vg, EdgeMatches* out) {
  *out = EdgeMatches();
  for (const auto& e : vg.GetAllEdges()) out->edges.push_back(e.first);
  std::sort(out->edges.begin(), out->edges.end());
  const size_t E = out->edges.size();
  std::unordered_map<theia::ViewIdPair, size_t> slot;
  slot.reserve(E * 2);
  for (size_t e = 0; e < E; ++e) slot[out->edges[e]] = e;

  // two passes over the tracks (count, fill): every pair of observations of one track whose views share an edge is a match
  std::vector<uint64_t> count(E + 1, 0);
  auto for_each_match = [&](auto&& fn) {
    for (const auto& t : tr.tracks)
      for (size_t a = 0; a < t.size(); ++a)
        for (size_t b = a + 1; b < t.size(); ++b) {
          if (t[a].first == t[b].first) continue;
          const bool swap = t[a].first > t[b].first;
          const auto& lo = swap ? t[b] : t[a];
          const auto& hi = swap ? t[a] : t[b];
          const auto it = slot.find(theia::ViewIdPair(lo.first, hi.first));
          if (it != slot.end()) fn(it->second, lo, hi);
        }
  };
  for_each_match([&](size_t e, const std::pair<theia::ViewId, int>&, const std::pair<theia::ViewId, int>&) { ++count[e + 1]; });
  out->match_ptr.assign(E + 1, 0);
  for (size_t e = 0; e < E; ++e) out->match_ptr[e + 1] = out->match_ptr[e] + count[e + 1];
  out->matches.assign(4 * out->match_ptr[E], 0.0);
  std::vector<uint64_t> cursor(out->match_ptr.begin(), out->match_ptr.end() - 1);
  for_each_match([&](size_t e, const std::pair<theia::ViewId, int>& lo, const std::pair<theia::ViewId, int>& hi) {
    const Eigen::Vector2d& p1 = tr.keypoints.at(lo.first)[lo.second];
    const Eigen::Vector2d& p2 = tr.keypoints.at(hi.first)[hi.second];
    double* m = &out->matches[4 * cursor[e]++];
    m[0] = p1[0]; m[1] = p1[1]; m[2] = p2[0]; m[3] = p2[1];
  });

  out->intrinsics.assign(6 * E, 0.0);
  out->rotation.assign(3 * E, 0.0);
  out->position.assign(3 * E, 0.0);
  for (size_t e = 0; e < E; ++e) {
    const theia::TwoViewInfo& info = *vg.GetEdge(out->edges[e].first, out->edges[e].second);
    const theia::ViewId v[2] = {out->edges[e].first, out->edges[e].second};
    for (int s = 0; s < 2; ++s) {
      const auto pp = tr.principal_point.find(v[s]);
      const auto fo = tr.focal.find(v[s]);
      const double u = pp == tr.principal_point.end() ? 0.0 : pp->second[0], w = pp == tr.principal_point.end() ? 0.0 : pp->second[1];
      // :99-104 reads CameraIntrinsicsPrior().focal_length, which is 0 without an EXIF entry (a singular K); the
      // median-viewing-angle guess the reader itself uses for TwoViewInfo (read_1dsfm.cc:347-357) stands in for it.
      double f = fo == tr.focal.end() ? 0.0 : fo->second;
      if (f == 0.0) f = 1.2 * u;
      out->intrinsics[6 * e + 3 * s + 0] = f;
      out->intrinsics[6 * e + 3 * s + 1] = u;
      out->intrinsics[6 * e + 3 * s + 2] = w;
    }
    for (int k = 0; k < 3; ++k) { out->rotation[3 * e + k] = info.rotation_2[k]; out->position[3 * e + k] = info.position_2[k]; }
  }
}

bool StoreCovarianceRot(const std::string& dir, const EdgeMatches& all, const theia::ViewGraph& vg, CovarianceMap* cov_out,
                        CalcCovarianceStats* stats, std::string* error) {
  // keep the pairs that are (still) edges of the view graph; rotation_2 / position_2 come from the graph's TwoViewInfo
  EdgeMatches em;
  em.match_ptr.push_back(0);
  for (size_t e = 0; e < all.edges.size(); ++e) {
    const theia::TwoViewInfo* info = vg.GetEdge(all.edges[e].first, all.edges[e].second);
    if (!info) continue;
    em.edges.push_back(all.edges[e]);
    em.matches.insert(em.matches.end(), all.matches.begin() + 4 * all.match_ptr[e], all.matches.begin() + 4 * all.match_ptr[e + 1]);
    em.match_ptr.push_back(em.matches.size() / 4);
    em.intrinsics.insert(em.intrinsics.end(), all.intrinsics.begin() + 6 * e, all.intrinsics.begin() + 6 * e + 6);
    for (int k = 0; k < 3; ++k) { em.rotation.push_back(info->rotation_2[k]); em.position.push_back(info->position_2[k]); }
  }
  const size_t E = em.edges.size();
  std::vector<double> cov9(9 * E), rot(3 * E), trans(3 * E);
  std::vector<int32_t> status(E), iters(E);
  double kernel_ms = 0.0;
  if (E > 0) {
    // 500 iterations: the reference's hard-coded Solver::Options (src/uncertainty.cpp:141-146)
    const gsfm_status st = gsfm_cov_estimate(E, em.match_ptr.data(), em.matches.data(), em.intrinsics.data(), em.rotation.data(), em.position.data(),
                                             500, cov9.data(), rot.data(), trans.data(), status.data(), iters.data(), &kernel_ms);
    if (st != GSFM_OK) { if (error) *error = gsfm_last_error(); return false; }
  }
  CovarianceMap result;
  CalcCovarianceStats s;
  s.num_edges = E;
  s.num_matches = em.match_ptr.back();
  s.kernel_ms = kernel_ms;
  for (size_t e = 0; e < E; ++e) {
    if (status[e] == 1) { ++s.num_skipped; continue; }
    if (status[e] != 0) { ++s.num_singular; continue; }
    Eigen::Matrix3d C;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = cov9[9 * e + 3 * r + c];
    result[em.edges[e]] = std::make_pair(C, Eigen::Vector3d(rot[3 * e], rot[3 * e + 1], rot[3 * e + 2]));
    ++s.num_written;
  }
  if (!WriteCovariance(dir, result)) { if (error) *error = "cannot write " + dir + "/covariance_rot.txt"; return false; }
  if (cov_out) *cov_out = std::move(result);
  if (stats) *stats = s;
  return true;
}

bool CalcCovariance(const std::string& dir, CovarianceMap* cov_out, CalcCovarianceStats* stats, std::string* error) {
  Tracks1DSfM tracks;
  if (!Read1DSFMTracks(dir, &tracks, error)) return false;
  theia::ViewGraph vg;
  if (!Read1DSFMViewGraph(dir, &vg, error)) return false;
  EdgeMatches em;
  CollectEdgeMatches(tracks, vg, &em);
  return StoreCovarianceRot(dir, em, vg, cov_out, stats, error);
}

// ------------------------------------------------------------------------------------------------------------------
// COLMAP export (two_views.txt)
// ------------------------------------------------------------------------------------------------------------------
bool ReadImageSize(const std::string& path, int* width, int* height) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return false;
  unsigned char h[26];
  f.read((char*)h, 24);
  if (f.gcount() >= 24 && h[0] == 0x89 && h[1] == 'P' && h[2] == 'N' && h[3] == 'G') {  // PNG: IHDR follows the 8-byte signature
    *width = (h[16] << 24) | (h[17] << 16) | (h[18] << 8) | h[19];
    *height = (h[20] << 24) | (h[21] << 16) | (h[22] << 8) | h[23];
    return true;
  }
  if (f.gcount() < 4 || h[0] != 0xFF || h[1] != 0xD8) return false;
  f.clear();
  f.seekg(2);
  for (;;) {  // JPEG: walk the marker segments to the first start-of-frame
    int c = f.get();
    if (c == EOF) return false;
    if (c != 0xFF) continue;
    int m = f.get();
    while (m == 0xFF) m = f.get();
    if (m == EOF) return false;
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;  // markers without a length
    unsigned char l[2];
    if (!f.read((char*)l, 2)) return false;
    const int len = (l[0] << 8) | l[1];
    const bool sof = m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC;
    if (sof) {
      unsigned char d[5];
      if (!f.read((char*)d, 5)) return false;
      *height = (d[1] << 8) | d[2];
      *width = (d[3] << 8) | d[4];
      return true;
    }
    f.seekg(len - 2, std::ios::cur);
  }
}

bool ReadColmapTwoViews(const std::string& path, const std::vector<std::string>& image_paths, ColmapPoseGraph* out, std::string* error) {
  *out = ColmapPoseGraph();
  std::unordered_map<std::string, Eigen::Vector2d> pp_by_name;
  for (const std::string& ip : image_paths) {
    int w = 0, h = 0;
    if (!ReadImageSize(ip, &w, &h)) { if (error) *error = "cannot read the size of image " + ip; return false; }
    const size_t slash = ip.find_last_of('/');
    pp_by_name[slash == std::string::npos ? ip : ip.substr(slash + 1)] = Eigen::Vector2d(w / 2, h / 2);  // integer halves, :80-82
  }
  std::ifstream fin(path);
  if (!fin.is_open()) { if (error) *error = "cannot read " + path; return false; }
  std::string line;
  for (int k = 0; k < 3; ++k) std::getline(fin, line);
  std::unordered_map<std::string, theia::ViewId> id_of;
  auto view_id = [&](const std::string& name) {
    auto it = id_of.find(name);
    if (it != id_of.end()) return it->second;
    const theia::ViewId id = (theia::ViewId)out->view_names.size();
    id_of[name] = id;
    out->view_names.push_back(name);
    const auto pp = pp_by_name.find(name);
    out->principal_point[id] = pp == pp_by_name.end() ? Eigen::Vector2d(0, 0) : pp->second;  // unknown image: operator[] default (:117-120)
    return id;
  };
  struct Pair { theia::ViewIdPair key; std::vector<double> m; double intr[6]; };
  std::vector<Pair> pairs;
  std::string n1, n2;
  double f1, f2, r[3], t[3];
  long num = 0;
  while (fin >> n1 >> n2 >> f1 >> f2 >> num >> r[0] >> r[1] >> r[2] >> t[0] >> t[1] >> t[2]) {
    if (num < 0) { if (error) *error = "negative inlier count in " + path; return false; }
    std::vector<double> a(2 * num), b(2 * num);
    for (double& v : a) if (!(fin >> v)) { if (error) *error = "two_views.txt ends inside a feature list"; return false; }
    for (double& v : b) if (!(fin >> v)) { if (error) *error = "two_views.txt ends inside a feature list"; return false; }
    theia::ViewId i = view_id(n1), j = view_id(n2);
    if (i == j) continue;
    theia::TwoViewInfo info;
    info.focal_length_1 = f1; info.focal_length_2 = f2;
    info.num_homography_inliers = info.num_verified_matches = info.visibility_score = (int)num;
    info.rotation_2 = Eigen::Vector3d(r[0], r[1], r[2]);
    info.position_2 = Eigen::Vector3d(t[0], t[1], t[2]);
    if (i > j) {  // SwapCameras: focal lengths swap, position <- -(R * position), rotation <- -rotation
      std::swap(info.focal_length_1, info.focal_length_2);
      double R[9], q[3];
      AngleAxisToRowMajorMatrix(info.rotation_2.data(), R);
      for (int a3 = 0; a3 < 3; ++a3) q[a3] = -(R[3 * a3] * t[0] + R[3 * a3 + 1] * t[1] + R[3 * a3 + 2] * t[2]);
      info.position_2 = Eigen::Vector3d(q[0], q[1], q[2]);
      info.rotation_2 = Eigen::Vector3d(-r[0], -r[1], -r[2]);
      std::swap(i, j);
      a.swap(b);
    }
    out->view_graph.AddEdge(i, j, info);
    Pair p;
    p.key = theia::ViewIdPair(i, j);
    p.m.resize(4 * num);
    for (long k = 0; k < num; ++k) { p.m[4 * k] = a[2 * k]; p.m[4 * k + 1] = a[2 * k + 1]; p.m[4 * k + 2] = b[2 * k]; p.m[4 * k + 3] = b[2 * k + 1]; }
    const Eigen::Vector2d &pi = out->principal_point[i], &pj = out->principal_point[j];
    const double intr[6] = {info.focal_length_1, pi[0], pi[1], info.focal_length_2, pj[0], pj[1]};
    std::memcpy(p.intr, intr, sizeof(intr));
    pairs.push_back(std::move(p));
  }
  // a pair listed twice: the later row wins, as AddEdge overwrites
  std::stable_sort(pairs.begin(), pairs.end(), [](const Pair& x, const Pair& y) { return x.key < y.key; });
  EdgeMatches& em = out->matches;
  em.match_ptr.push_back(0);
  for (size_t k = 0; k < pairs.size(); ++k) {
    if (k + 1 < pairs.size() && pairs[k + 1].key == pairs[k].key) continue;
    const theia::TwoViewInfo& info = *out->view_graph.GetEdge(pairs[k].key.first, pairs[k].key.second);
    em.edges.push_back(pairs[k].key);
    em.matches.insert(em.matches.end(), pairs[k].m.begin(), pairs[k].m.end());
    em.match_ptr.push_back(em.matches.size() / 4);
    em.intrinsics.insert(em.intrinsics.end(), pairs[k].intr, pairs[k].intr + 6);
    for (int c = 0; c < 3; ++c) { em.rotation.push_back(info.rotation_2[c]); em.position.push_back(info.position_2[c]); }
  }
  return true;
}

std::vector<std::string> ExpandWildcard(const std::string& pattern) {
  std::vector<std::string> out;
  glob_t g;
  if (glob(pattern.c_str(), 0, nullptr, &g) == 0) {
    for (size_t k = 0; k < g.gl_pathc; ++k) out.emplace_back(g.gl_pathv[k]);
  }
  globfree(&g);
  return out;
}

}  // namespace gsfm
