// The callers and data formats either side of the hot path ("next" rows of SURVEY section 8f),
// host-side C++: a minimal theia::ViewGraph, the maximum-spanning-tree initialisation, the 1DSfM
// EGs.txt / covariance_rot.txt codecs, the post-rotation edge filter and the evaluation metrics.
#pragma once
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "compat.hpp"

namespace theia {

// Subset of thirdparty/TheiaSfM/src/theia/sfm/view_graph/view_graph.h used around the rotation path.
class ViewGraph {
 public:
  int NumViews() const { return (int)vertices_.size(); }
  int NumEdges() const { return (int)edges_.size(); }
  bool HasView(ViewId v) const { return vertices_.count(v) != 0; }
  bool HasEdge(ViewId a, ViewId b) const { return edges_.count(Key(a, b)) != 0; }
  std::unordered_set<ViewId> ViewIds() const;
  // The key is normalised to (min, max); the payload is stored untouched (view_graph.cc:133-153).
  void AddEdge(ViewId a, ViewId b, const TwoViewInfo& info);
  bool RemoveEdge(ViewId a, ViewId b);
  bool RemoveView(ViewId v);
  const TwoViewInfo* GetEdge(ViewId a, ViewId b) const;
  const std::unordered_set<ViewId>* GetNeighborIdsForView(ViewId v) const;
  const std::unordered_map<ViewIdPair, TwoViewInfo>& GetAllEdges() const { return edges_; }
  void GetLargestConnectedComponentIds(std::unordered_set<ViewId>* out) const;
  void ExtractSubgraph(const std::unordered_set<ViewId>& keep, ViewGraph* sub) const;
  static ViewIdPair Key(ViewId a, ViewId b) { return a < b ? ViewIdPair(a, b) : ViewIdPair(b, a); }

 private:
  std::unordered_map<ViewId, std::unordered_set<ViewId>> vertices_;
  std::unordered_map<ViewIdPair, TwoViewInfo> edges_;
};

#ifndef GSFM_USE_REAL_THEIA
// Only what the rotation stage reads/writes of a theia::Reconstruction: the set of views and the
// estimated orientation per view (SetOrientations, bind_src/GlobalSfMpy.cpp:80-98). Tracks, cameras and
// intrinsics are out of scope.
class Reconstruction {
 public:
  std::unordered_map<ViewId, Eigen::Vector3d> orientation;
  std::unordered_set<ViewId> views;
  int NumTracks() const { return 0; }
  int NumViews() const { return (int)views.size(); }
};
#endif

// thirdparty/TheiaSfM/src/theia/sfm/view_graph/orientations_from_maximum_spanning_tree.cc:109-181
bool OrientationsFromMaximumSpanningTree(const ViewGraph& view_graph, std::unordered_map<ViewId, Eigen::Vector3d>* orientations);

// thirdparty/TheiaSfM/src/theia/sfm/filter_view_pairs_from_orientation.cc:55-122: drops the edges whose
// relative rotation disagrees with the global orientations by more than the threshold.
void FilterViewPairsFromOrientation(const std::unordered_map<ViewId, Eigen::Vector3d>& orientations,
                                    double max_relative_rotation_difference_degrees, ViewGraph* view_graph);
// thirdparty/TheiaSfM/src/theia/sfm/view_graph/remove_disconnected_view_pairs.cc: keeps the largest component.
std::unordered_set<ViewId> RemoveDisconnectedViewPairs(ViewGraph* view_graph);

}  // namespace theia

namespace gsfm {
// 1DSfM: EGs.txt (+ cc.txt restriction) -> view graph. Convention of io/read_1dsfm.cc:299-372:
// R' = S R^T S with S = diag(1,-1,-1), no re-orthonormalisation, t' = S t.  tracks.txt/coords.txt are not
// read (camera priors and tracks are out of scope); num_verified_matches stays 0 unless `matches` is given.
bool Read1DSFMViewGraph(const std::string& dataset_directory, theia::ViewGraph* view_graph, std::string* error);
// covariance_rot.txt codec (src/uncertainty.cpp:164-230): doubles stored as the decimal of their bit pattern.
bool ReadCovariance(const std::string& dataset_directory, CovarianceMap* covariances);
bool WriteCovariance(const std::string& dataset_directory, const CovarianceMap& covariances);
// Per-edge angular residual || log(R_ij^T R_j R_i^T) || in degrees (src/compare_reconstructions.cpp:617-647).
std::vector<double> ResidualsOfRelativeRotations(const theia::ViewGraph& view_graph,
                                                 const std::unordered_map<theia::ViewId, Eigen::Vector3d>& orientations);
}  // namespace gsfm
