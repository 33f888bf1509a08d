// The callers and data formats either side of the hot path ("next" rows of SURVEY section 8f),
// host-side C++: a minimal theia::ViewGraph, the maximum-spanning-tree initialisation, the 1DSfM
// EGs.txt / covariance_rot.txt codecs, the post-rotation edge filter and the evaluation metrics.
#pragma once
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <memory>

#include "compat.hpp"

namespace gsfm { struct EdgeMatches; }

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
  std::unordered_map<ViewId, std::string> view_names;  // COLMAP ingestion: image file name per view
  // The matched features of the view pairs, in place of theia's tracks: all that store_covariance_rot reads of them
  // (get_matched_features, src/uncertainty.cpp:3-33).  Set by Read1DSFM (when tracks.txt exists) and by
  // AddColmapMatchesToReconstructionBuilder.
  std::shared_ptr<const gsfm::EdgeMatches> matches;
  int NumTracks() const { return 0; }
  int NumViews() const { return (int)views.size(); }
};
#endif

// thirdparty/TheiaSfM/src/theia/sfm/view_graph/orientations_from_maximum_spanning_tree.cc:109-181
bool OrientationsFromMaximumSpanningTree(const ViewGraph& view_graph, std::unordered_map<ViewId, Eigen::Vector3d>* orientations);

// thirdparty/TheiaSfM/src/theia/sfm/filter_view_pairs_from_orientation.cc:55-122: drops the edges whose
// relative rotation disagrees with the global orientations by more than the threshold (and the edges touching a view
// without an orientation).  The angles of all edges are one device sweep (gsfm_rot_edge_sq_norms); throws
// std::runtime_error when no device is usable (there is no CPU fallback).
void FilterViewPairsFromOrientation(const std::unordered_map<ViewId, Eigen::Vector3d>& orientations,
                                    double max_relative_rotation_difference_degrees, ViewGraph* view_graph);
// thirdparty/TheiaSfM/src/theia/sfm/view_graph/remove_disconnected_view_pairs.cc: keeps the largest component.
std::unordered_set<ViewId> RemoveDisconnectedViewPairs(ViewGraph* view_graph);

}  // namespace theia

namespace gsfm {
// 1DSfM: EGs.txt (+ cc.txt restriction) -> view graph. Convention of io/read_1dsfm.cc:299-372:
// R' = S R^T S with S = diag(1,-1,-1), no re-orthonormalisation, t' = S t.  When tracks.txt / coords.txt / list.txt are
// present, num_verified_matches = number of common tracks and the focal lengths follow read_1dsfm.cc:341-362.
bool Read1DSFMViewGraph(const std::string& dataset_directory, theia::ViewGraph* view_graph, std::string* error);
void AnnotateViewGraphFromTracks(const std::string& dataset_directory, theia::ViewGraph* view_graph);
// covariance_rot.txt codec (src/uncertainty.cpp:164-230): doubles stored as the decimal of their bit pattern.
bool ReadCovariance(const std::string& dataset_directory, CovarianceMap* covariances);
bool WriteCovariance(const std::string& dataset_directory, const CovarianceMap& covariances);

// ---- 1DSfM keypoints and tracks + the CalcCovariance driver ----
// What thirdparty/TheiaSfM/src/theia/io/read_1dsfm.cc:114-292 reads besides the epipolar geometries: list.txt (view id =
// line index, optional EXIF focal), coords.txt (per view: principal point, keypoints) and tracks.txt.
struct Tracks1DSfM {
  std::unordered_map<theia::ViewId, double> focal;  // list.txt; 0 = no EXIF focal (read_1dsfm.cc:136-142)
  std::unordered_map<theia::ViewId, std::string> names;  // list.txt: file name without its directory (:128-130)
  std::unordered_map<theia::ViewId, Eigen::Vector2d> principal_point;  // coords.txt header (parsed as float, :176-193)
  std::unordered_map<theia::ViewId, std::vector<Eigen::Vector2d>> keypoints;
  std::vector<std::vector<std::pair<theia::ViewId, int>>> tracks;  // (view, keypoint index)
};
bool Read1DSFMTracks(const std::string& dataset_directory, Tracks1DSfM* out, std::string* error);

// The matched features of every view-graph edge, flattened for gsfm_cov_estimate: the common tracks of the two views
// (get_matched_features, src/uncertainty.cpp:3-33), the intrinsics of :99-104 and TwoViewInfo::rotation_2/position_2.
struct EdgeMatches {
  std::vector<theia::ViewIdPair> edges;  // sorted by key
  std::vector<uint64_t> match_ptr;       // edges.size() + 1
  std::vector<double> matches;           // x1 y1 x2 y2
  std::vector<double> intrinsics;        // f1 u1 v1 f2 u2 v2
  std::vector<double> rotation, position;
};
void CollectEdgeMatches(const Tracks1DSfM& tracks, const theia::ViewGraph& view_graph, EdgeMatches* out);

// two_views.txt of scripts/read_colmap_database.py:52-133 as AddColmapMatchesToReconstructionBuilder reads it
// (src/read_colmap_posegraph.cpp:55-164): three header lines, then per pair
//   "<image1> <image2> f1 f2 num_inliers rot[3] trans[3]" / num_inliers "x y" of image 1 / num_inliers "x y" of image 2.
// View ids follow first appearance (the builder's AddImage order); a pair listed larger-id-first is swapped to
// smaller->larger like GSfMReconstructionBuilder::AddMatchToViewGraph does (SwapCameras, twoview_info.cc:44-62).
// The principal point is half the image size (:77-85); sizes are read from the JPEG/PNG headers of `image_paths`.
// Deviation: the reference pushes the correspondences through theia's TrackBuilder and later recovers an edge's matches
// from the common tracks; here an edge keeps exactly the inlier correspondences listed for it.
struct ColmapPoseGraph {
  std::vector<std::string> view_names;  // index = view id
  std::unordered_map<theia::ViewId, Eigen::Vector2d> principal_point;
  theia::ViewGraph view_graph;
  EdgeMatches matches;
};
bool ReadColmapTwoViews(const std::string& two_views_path, const std::vector<std::string>& image_paths, ColmapPoseGraph* out, std::string* error);
bool ReadImageSize(const std::string& path, int* width, int* height);  // JPEG (SOFn) and PNG (IHDR) headers
std::vector<std::string> ExpandWildcard(const std::string& pattern);

struct CalcCovarianceStats {
  size_t num_edges = 0, num_matches = 0, num_written = 0, num_skipped = 0, num_singular = 0;
  double kernel_ms = 0.0;
};
// bind_src/GlobalSfMpy.cpp:623-628 + store_covariance_rot (src/uncertainty.cpp:164-198): read the dataset, estimate the
// rotation covariance of every edge on the device (one wavefront per edge, gsfm_cov_estimate) and write
// covariance_rot.txt with the refined relative rotation beside it.  Edges the reference skips (zero translation, no
// common track, singular information matrix) are not written.
bool CalcCovariance(const std::string& dataset_directory, CovarianceMap* covariances_or_null, CalcCovarianceStats* stats,
                    std::string* error);
// store_covariance_rot (src/uncertainty.cpp:164-198) on already-flattened matches: the edges of `view_graph` that have
// matches are estimated in one device launch and written to <dir>/covariance_rot.txt.
bool StoreCovarianceRot(const std::string& dataset_directory, const EdgeMatches& matches, const theia::ViewGraph& view_graph,
                        CovarianceMap* covariances_or_null, CalcCovarianceStats* stats, std::string* error);

// residuals_of_relative_rot (src/compare_reconstructions.cpp:617-647): for every edge that has a covariance, the norm of the whitened
// residual Lt log(R_j R_i^T R_ij^T), Lt from 1e8 * Sigma_e, at the given orientations; edges in ViewIdPair order (the reference walks its
// unordered_map).  Edges whose views lack an orientation are skipped.  One device sweep (gsfm_rot_edge_sq_norms); throws without a device.
void ResidualsOfRelativeRotations(const theia::ViewGraph& view_graph, const std::unordered_map<theia::ViewId, Eigen::Vector3d>& orientations,
                                  const CovarianceMap& covariances, std::vector<double>* residuals);
}  // namespace gsfm
