// Evaluation metrics of the rotation stage, the counterparts of the reference's
// include/compare_reconstructions.hpp and include/read_colmap_posegraph.hpp restricted to orientations.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "view_graph.hpp"

namespace gsfm {

// Angle of R1^T R2 in radians, in [0, pi]   (src/compare_reconstructions.cpp:7-16)
double AngularDifference(const Eigen::Vector3d& rotation1, const Eigen::Vector3d& rotation2);

struct AlignmentSummary {
  Eigen::Vector3d alignment;  // angle-axis a: every rotation was replaced by R_i * Exp(a)
  double initial_cost = 0.0, final_cost = 0.0;
  int iterations = 0;
  bool converged = false;
  std::string message;
};
// Robust global alignment (src/compare_reconstructions.cpp:147-177): argmin_a sum_i Cauchy(0.1)(|gt_i - Log(R_i Exp(a))|^2),
// Ceres LM defaults with 500 iterations and function_tolerance 0, then R_i <- R_i Exp(a) in place.
AlignmentSummary AlignRotations(const std::vector<Eigen::Vector3d>& gt_rotation, std::vector<Eigen::Vector3d>* rotation);

// COLMAP images.txt poses (src/read_colmap_posegraph.cpp:5-53): pose = {tx, ty, tz, angle-axis of (qw, qx, qy, qz)}.
struct ColmapViewGraph {
  void read_poses(const std::string& path);
  int num_view = 0;
  std::unordered_map<std::string, uint32_t> image_ids;
  std::unordered_map<uint32_t, std::string> image_names;
  std::unordered_map<uint32_t, std::vector<double>> poses;
};

struct CompareInfo {  // include/compare_reconstructions.hpp; only the orientation fields are ever filled here
  std::vector<double> rotation_diff_when_align;  // radians per common view, after AlignRotations
  std::vector<double> position_errors;
  int num_3d_points = 0;
  int common_camera = 0;
  int num_reconstructed_view = 0;
};

// Views are matched by name (theia::Reconstruction::view_names); "estimated" = has an orientation.
std::vector<std::string> FindCommonEstimatedViewsByName(const theia::Reconstruction& a, const theia::Reconstruction& b);
std::vector<std::string> FindCommonEstimatedViewsByNameColmap(const ColmapViewGraph& colmap, const theia::Reconstruction& reconstruction);
CompareInfo compare_orientations(const std::vector<std::string>& common_view_names, const theia::Reconstruction& reference,
                                 theia::Reconstruction* reconstruction_to_align, double robust_alignment_threshold);
CompareInfo compare_orientations_colmap(const std::vector<std::string>& common_view_names, const ColmapViewGraph& reference,
                                        theia::Reconstruction* reconstruction_to_align, double robust_alignment_threshold);

}  // namespace gsfm
