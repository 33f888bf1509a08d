// Host-side callers / data formats around the rotation path (SURVEY section 8f "next" rows).
#include "../../include/gsfm/view_graph.hpp"
#include "../../include/gsfm_rot.h"

#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <numeric>
#include <sstream>
#include <stdexcept>

namespace {

// ---- rotation helpers (ceres/rotation.h semantics, row-major 3x3) ----
void aa_to_matrix(const double* a, double* R) {
  const double t2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  if (t2 > std::numeric_limits<double>::epsilon()) {
    const double t = std::sqrt(t2), wx = a[0] / t, wy = a[1] / t, wz = a[2] / t, c = std::cos(t), s = std::sin(t), k = 1.0 - c;
    R[0] = c + wx * wx * k; R[1] = wx * wy * k - wz * s; R[2] = wy * s + wx * wz * k;
    R[3] = wz * s + wx * wy * k; R[4] = c + wy * wy * k; R[5] = -wx * s + wy * wz * k;
    R[6] = -wy * s + wx * wz * k; R[7] = wx * s + wy * wz * k; R[8] = c + wz * wz * k;
  } else {
    R[0] = 1; R[1] = -a[2]; R[2] = a[1]; R[3] = a[2]; R[4] = 1; R[5] = -a[0]; R[6] = -a[1]; R[7] = a[0]; R[8] = 1;
  }
}
void matrix_to_aa(const double* R, double* a) {
  double q[4];
  const double tr = R[0] + R[4] + R[8];
  if (tr >= 0.0) {
    double t = std::sqrt(tr + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i + 1] = 0.5 * t; t = 0.5 / t;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double s = std::sqrt(s2);
    const double tt = 2.0 * ((q[0] < 0.0) ? std::atan2(-s, -q[0]) : std::atan2(s, q[0]));
    const double k = tt / s;
    a[0] = q[1] * k; a[1] = q[2] * k; a[2] = q[3] * k;
  } else { a[0] = 2 * q[1]; a[1] = 2 * q[2]; a[2] = 2 * q[3]; }
}
void mul(const double* A, const double* B, double* C, bool transposeA, bool transposeB) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (transposeA ? A[3 * k + r] : A[3 * r + k]) * (transposeB ? B[3 * c + k] : B[3 * k + c]);
    C[3 * r + c] = s;
  }
}

struct DSU {
  std::vector<uint32_t> p;
  explicit DSU(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
  uint32_t find(uint32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
  bool unite(uint32_t a, uint32_t b) { a = find(a); b = find(b); if (a == b) return false; p[a] = b; return true; }
};

}  // namespace

namespace theia {

std::unordered_set<ViewId> ViewGraph::ViewIds() const {
  std::unordered_set<ViewId> out;
  for (const auto& kv : vertices_) out.insert(kv.first);
  return out;
}
void ViewGraph::AddEdge(ViewId a, ViewId b, const TwoViewInfo& info) {
  if (a == b) return;
  vertices_[a].insert(b);
  vertices_[b].insert(a);
  edges_[Key(a, b)] = info;
}
bool ViewGraph::RemoveEdge(ViewId a, ViewId b) {
  if (!edges_.erase(Key(a, b))) return false;
  vertices_[a].erase(b);
  vertices_[b].erase(a);
  return true;
}
bool ViewGraph::RemoveView(ViewId v) {
  auto it = vertices_.find(v);
  if (it == vertices_.end()) return false;
  for (ViewId n : it->second) { vertices_[n].erase(v); edges_.erase(Key(v, n)); }
  vertices_.erase(it);
  return true;
}
const TwoViewInfo* ViewGraph::GetEdge(ViewId a, ViewId b) const {
  auto it = edges_.find(Key(a, b));
  return it == edges_.end() ? nullptr : &it->second;
}
const std::unordered_set<ViewId>* ViewGraph::GetNeighborIdsForView(ViewId v) const {
  auto it = vertices_.find(v);
  return it == vertices_.end() ? nullptr : &it->second;
}
void ViewGraph::GetLargestConnectedComponentIds(std::unordered_set<ViewId>* out) const {
  out->clear();
  std::vector<ViewId> ids;
  for (const auto& kv : vertices_) ids.push_back(kv.first);
  std::sort(ids.begin(), ids.end());
  std::unordered_map<ViewId, uint32_t> idx;
  for (size_t k = 0; k < ids.size(); ++k) idx[ids[k]] = (uint32_t)k;
  DSU d(ids.size());
  for (const auto& e : edges_) d.unite(idx[e.first.first], idx[e.first.second]);
  std::unordered_map<uint32_t, uint32_t> count;
  for (size_t k = 0; k < ids.size(); ++k) count[d.find((uint32_t)k)]++;
  uint32_t best = 0, best_n = 0;
  for (size_t k = 0; k < ids.size(); ++k) {  // deterministic: first (smallest id) among the largest
    const uint32_t r = d.find((uint32_t)k);
    if (count[r] > best_n) { best_n = count[r]; best = r; }
  }
  for (size_t k = 0; k < ids.size(); ++k) if (d.find((uint32_t)k) == best) out->insert(ids[k]);
}
void ViewGraph::ExtractSubgraph(const std::unordered_set<ViewId>& keep, ViewGraph* sub) const {
  for (const auto& e : edges_)
    if (keep.count(e.first.first) && keep.count(e.first.second)) sub->AddEdge(e.first.first, e.first.second, e.second);
}

bool OrientationsFromMaximumSpanningTree(const ViewGraph& view_graph, std::unordered_map<ViewId, Eigen::Vector3d>* orientations) {
  if (!orientations) return false;
  std::unordered_set<ViewId> cc;
  view_graph.GetLargestConnectedComponentIds(&cc);  // :116-119
  if (cc.empty()) return false;
  struct E { int w; ViewId a, b; };
  std::vector<E> es;
  for (const auto& e : view_graph.GetAllEdges())
    if (cc.count(e.first.first) && cc.count(e.first.second)) es.push_back({e.second.num_verified_matches, e.first.first, e.first.second});
  // Kruskal on negated weights (:122-135); ties broken by the view ids so the tree is reproducible
  std::sort(es.begin(), es.end(), [](const E& x, const E& y) { return x.w != y.w ? x.w > y.w : (x.a != y.a ? x.a < y.a : x.b < y.b); });
  std::vector<ViewId> ids(cc.begin(), cc.end());
  std::sort(ids.begin(), ids.end());
  std::unordered_map<ViewId, uint32_t> idx;
  for (size_t k = 0; k < ids.size(); ++k) idx[ids[k]] = (uint32_t)k;
  DSU d(ids.size());
  std::unordered_map<ViewId, std::vector<std::pair<ViewId, int>>> tree;
  size_t used = 0;
  for (const E& e : es) if (d.unite(idx[e.a], idx[e.b])) { tree[e.a].push_back({e.b, e.w}); tree[e.b].push_back({e.a, e.w}); ++used; }
  if (used + 1 != ids.size()) return false;
  // walk the tree from the root with a heap on num_verified_matches (:146-178)
  struct H { int w; ViewId src, dst; };
  auto cmp = [](const H& x, const H& y) { return x.w != y.w ? x.w < y.w : (x.dst != y.dst ? x.dst > y.dst : x.src > y.src); };
  std::vector<H> heap;
  const ViewId root = ids.front();
  (*orientations)[root] = Eigen::Vector3d::Zero();
  auto push_edges = [&](ViewId v) {
    for (const auto& nb : tree[v]) if (!orientations->count(nb.first)) { heap.push_back({nb.second, v, nb.first}); std::push_heap(heap.begin(), heap.end(), cmp); }
  };
  push_edges(root);
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    const H h = heap.back();
    heap.pop_back();
    if (orientations->count(h.dst)) continue;
    double Rs[9], Rr[9], Rn[9], aa[3];
    aa_to_matrix(orientations->at(h.src).data(), Rs);
    aa_to_matrix(view_graph.GetEdge(h.src, h.dst)->rotation_2.data(), Rr);
    mul(Rr, Rs, Rn, /*transposeA=*/!(h.src < h.dst), false);  // R_nbr = R_rel R_src, or R_rel^T R_src (:73-76)
    matrix_to_aa(Rn, aa);
    (*orientations)[h.dst] = Eigen::Vector3d(aa[0], aa[1], aa[2]);
    push_edges(h.dst);
  }
  return true;
}

namespace {
// edges whose two views have an orientation, in ViewIdPair order, flattened for the C-ABI (dense camera index = rank of the ViewId)
struct FlatEdges {
  std::vector<ViewIdPair> keys;
  std::vector<uint32_t> ei, ej;
  std::vector<double> rel, rot;
  uint32_t n_cams = 0;
};
FlatEdges flatten_edges(const ViewGraph& vg, const std::unordered_map<ViewId, Eigen::Vector3d>& orientations, std::vector<ViewIdPair>* without_orientation) {
  FlatEdges f;
  std::vector<ViewId> ids;
  for (const auto& kv : orientations) ids.push_back(kv.first);
  std::sort(ids.begin(), ids.end());
  std::unordered_map<ViewId, uint32_t> index;
  for (size_t k = 0; k < ids.size(); ++k) index[ids[k]] = (uint32_t)k;
  f.n_cams = (uint32_t)ids.size();
  f.rot.resize(3 * ids.size());
  for (size_t k = 0; k < ids.size(); ++k) { const Eigen::Vector3d& w = orientations.at(ids[k]); f.rot[3 * k] = w[0]; f.rot[3 * k + 1] = w[1]; f.rot[3 * k + 2] = w[2]; }
  std::vector<ViewIdPair> keys;
  for (const auto& e : vg.GetAllEdges()) keys.push_back(e.first);
  std::sort(keys.begin(), keys.end());
  for (const ViewIdPair& k : keys) {
    auto i1 = index.find(k.first), i2 = index.find(k.second);
    if (i1 == index.end() || i2 == index.end()) { if (without_orientation) without_orientation->push_back(k); continue; }
    const Eigen::Vector3d& r = vg.GetEdge(k.first, k.second)->rotation_2;
    f.keys.push_back(k); f.ei.push_back(i1->second); f.ej.push_back(i2->second);
    f.rel.push_back(r[0]); f.rel.push_back(r[1]); f.rel.push_back(r[2]);
  }
  return f;
}
}  // namespace

void FilterViewPairsFromOrientation(const std::unordered_map<ViewId, Eigen::Vector3d>& orientations,
                                    double max_deg, ViewGraph* view_graph) {
  const double thr = max_deg * M_PI / 180.0, thr2 = thr * thr;
  std::vector<ViewIdPair> bad;   // a view pair with a view that has no orientation is removed (:92-101)
  const FlatEdges f = flatten_edges(*view_graph, orientations, &bad);
  if (!f.keys.empty()) {
    std::vector<double> s(f.keys.size());
    std::vector<uint8_t> keep(f.keys.size());
    uint64_t kept = 0;
    // loop = R_rel^T (R_2 R_1^T), squared angle against the squared threshold (:60-66), all edges in one device sweep
    const gsfm_status st = gsfm_rot_edge_sq_norms(f.n_cams, f.keys.size(), f.ei.data(), f.ej.data(), f.rel.data(), nullptr, f.rot.data(), thr2,
                                                  s.data(), keep.data(), &kept, nullptr);
    if (st != GSFM_OK) throw std::runtime_error(std::string("FilterViewPairsFromOrientation: ") + gsfm_last_error());
    for (size_t e = 0; e < f.keys.size(); ++e) if (!keep[e]) bad.push_back(f.keys[e]);
  }
  for (const auto& k : bad) view_graph->RemoveEdge(k.first, k.second);
}

std::unordered_set<ViewId> RemoveDisconnectedViewPairs(ViewGraph* view_graph) {
  std::unordered_set<ViewId> keep, removed;
  view_graph->GetLargestConnectedComponentIds(&keep);
  for (ViewId v : view_graph->ViewIds()) if (!keep.count(v)) removed.insert(v);
  for (ViewId v : removed) view_graph->RemoveView(v);
  return removed;
}

}  // namespace theia

namespace gsfm {

bool Read1DSFMViewGraph(const std::string& dir, theia::ViewGraph* view_graph, std::string* error) {
  std::unordered_set<theia::ViewId> cc;
  {
    std::ifstream f(dir + "/cc.txt");  // read_1dsfm.cc:93-109: only views of the largest component file
    if (f.is_open()) { theia::ViewId v; while (f >> v) cc.insert(v); }
  }
  std::ifstream ifs(dir + "/EGs.txt");
  if (!ifs.is_open()) { if (error) *error = "cannot read " + dir + "/EGs.txt"; return false; }
  std::string line;
  while (std::getline(ifs, line)) {
    std::istringstream ss(line);
    theia::ViewId a, b;
    double R[9], t[3];
    if (!(ss >> a >> b)) continue;
    bool ok = true;
    for (int k = 0; k < 9; ++k) ok &= (bool)(ss >> R[k]);
    for (int k = 0; k < 3; ++k) ok &= (bool)(ss >> t[k]);
    if (!ok) { if (error) *error = "malformed EGs.txt line"; return false; }
    if (!cc.empty() && (!cc.count(a) || !cc.count(b))) continue;
    // R' = S R^T S, S = diag(1,-1,-1)   (read_1dsfm.cc:307-325)
    const double S[3] = {1.0, -1.0, -1.0};
    double Rp[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rp[3 * r + c] = S[r] * R[3 * c + r] * S[c];
    theia::TwoViewInfo info;
    matrix_to_aa(Rp, info.rotation_2.data());
    for (int k = 0; k < 3; ++k) info.position_2[k] = S[k] * t[k];
    view_graph->AddEdge(a, b, info);
  }
  AnnotateViewGraphFromTracks(dir, view_graph);
  return true;
}

// read_1dsfm.cc:341-368: focal lengths from the EXIF prior (else 1.2 * principal point x) and
// num_verified_matches = visibility_score = number of common tracks -- when the dataset carries tracks at all.
void AnnotateViewGraphFromTracks(const std::string& dir, theia::ViewGraph* view_graph) {
  for (const char* name : {"/list.txt", "/coords.txt", "/tracks.txt"})
    if (!std::ifstream(dir + name).is_open()) return;
  Tracks1DSfM tracks;
  if (!Read1DSFMTracks(dir, &tracks, nullptr)) return;
  EdgeMatches em;
  CollectEdgeMatches(tracks, *view_graph, &em);
  for (size_t e = 0; e < em.edges.size(); ++e) {
    theia::TwoViewInfo info = *view_graph->GetEdge(em.edges[e].first, em.edges[e].second);
    info.focal_length_1 = em.intrinsics[6 * e];
    info.focal_length_2 = em.intrinsics[6 * e + 3];
    info.num_verified_matches = info.visibility_score = (int)(em.match_ptr[e + 1] - em.match_ptr[e]);
    view_graph->AddEdge(em.edges[e].first, em.edges[e].second, info);
  }
}

bool ReadCovariance(const std::string& dir, CovarianceMap* covariances) {
  std::ifstream f(dir + "/covariance_rot.txt");
  if (!f.is_open()) return false;
  covariances->clear();
  std::string line;
  std::getline(f, line);
  std::getline(f, line);  // two header lines (uncertainty.cpp:208-209)
  while (std::getline(f, line)) {
    unsigned id1, id2;
    uint64_t u[9];
    if (std::sscanf(line.c_str(), "%u %u %" SCNu64 " %" SCNu64 " %" SCNu64 " %" SCNu64 " %" SCNu64 " %" SCNu64 " %" SCNu64 " %" SCNu64 " %" SCNu64,
                    &id1, &id2, &u[0], &u[1], &u[2], &u[3], &u[4], &u[5], &u[6], &u[7], &u[8]) != 11) continue;
    double d[9];
    std::memcpy(d, u, sizeof(d));
    Eigen::Matrix3d C;  // C00 C11 C22 C01 C02 C12 (:219-222)
    C(0, 0) = d[0]; C(1, 1) = d[1]; C(2, 2) = d[2];
    C(0, 1) = C(1, 0) = d[3]; C(0, 2) = C(2, 0) = d[4]; C(1, 2) = C(2, 1) = d[5];
    (*covariances)[theia::ViewIdPair(id1, id2)] = std::make_pair(C, Eigen::Vector3d(d[6], d[7], d[8]));
  }
  return true;
}

bool WriteCovariance(const std::string& dir, const CovarianceMap& covariances) {
  std::ofstream out(dir + "/covariance_rot.txt");
  if (!out.is_open()) return false;
  out << "# Stored as uint64, should convert to double first.\n# view_id1 view_id2 C00 C11 C22 C01 C02 C12 R0 R1 R2" << std::endl;
  std::vector<theia::ViewIdPair> keys;
  for (const auto& kv : covariances) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());
  for (const auto& k : keys) {
    const auto& v = covariances.at(k);
    const double d[9] = {v.first(0, 0), v.first(1, 1), v.first(2, 2), v.first(0, 1), v.first(0, 2), v.first(1, 2), v.second[0], v.second[1], v.second[2]};
    uint64_t u[9];
    std::memcpy(u, d, sizeof(u));
    out << k.first << " " << k.second;
    for (int c = 0; c < 9; ++c) out << " " << u[c];
    out << " " << std::endl;
  }
  return true;
}

void ResidualsOfRelativeRotations(const theia::ViewGraph& view_graph, const std::unordered_map<theia::ViewId, Eigen::Vector3d>& orientations,
                                  const CovarianceMap& covariances, std::vector<double>* residuals) {
  residuals->clear();
  theia::FlatEdges f = theia::flatten_edges(view_graph, orientations, nullptr);
  // only the edges that have a covariance (:631)
  std::vector<uint32_t> ei, ej; std::vector<double> rel, cov6;
  for (size_t e = 0; e < f.keys.size(); ++e) {
    auto it = covariances.find(f.keys[e]);
    if (it == covariances.end()) continue;
    const Eigen::Matrix3d& C = it->second.first;
    ei.push_back(f.ei[e]); ej.push_back(f.ej[e]);
    for (int c = 0; c < 3; ++c) rel.push_back(f.rel[3 * e + c]);
    const double c6[6] = {C(0, 0), C(1, 1), C(2, 2), C(0, 1), C(0, 2), C(1, 2)};
    cov6.insert(cov6.end(), c6, c6 + 6);
  }
  if (ei.empty()) return;
  residuals->resize(ei.size());
  const gsfm_status st = gsfm_rot_edge_sq_norms(f.n_cams, ei.size(), ei.data(), ej.data(), rel.data(), cov6.data(), f.rot.data(), -1.0,
                                                residuals->data(), nullptr, nullptr, nullptr);
  if (st != GSFM_OK) throw std::runtime_error(std::string("residuals_of_relative_rot: ") + gsfm_last_error());
  for (double& v : *residuals) v = std::sqrt(v);   // sqrt(r0^2 + r1^2 + r2^2) (:643)
}

}  // namespace gsfm
