// 1DSfM keypoints/tracks ingestion and the CalcCovariance driver: the caller of the batched per-edge covariance kernel
// (SURVEY 8f rows 2 and 4).  Follows thirdparty/TheiaSfM/src/theia/io/read_1dsfm.cc:93-292 for the file formats and
// src/uncertainty.cpp:3-33,82-198 + bind_src/GlobalSfMpy.cpp:623-628 for what is estimated and written.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../../include/gsfm/view_graph.hpp"
#include "../../include/gsfm_rot.h"

namespace gsfm {

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
      if (cc.empty() || cc.count(id)) out->focal[id] = focal;
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

bool CalcCovariance(const std::string& dir, CovarianceMap* cov_out, CalcCovarianceStats* stats, std::string* error) {
  Tracks1DSfM tracks;
  if (!Read1DSFMTracks(dir, &tracks, error)) return false;
  theia::ViewGraph vg;
  if (!Read1DSFMViewGraph(dir, &vg, error)) return false;
  EdgeMatches em;
  CollectEdgeMatches(tracks, vg, &em);
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
  s.num_matches = em.match_ptr.empty() ? 0 : em.match_ptr.back();
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

}  // namespace gsfm
