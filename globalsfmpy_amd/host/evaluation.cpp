// Evaluation metrics of the rotation stage: AngularDifference, the robust AlignRotations, and the
// compare_orientations family with the COLMAP images.txt reader -- host-side C++ like the reference's
// src/compare_reconstructions.cpp:7-16,140-177,197-259,262-296 and src/read_colmap_posegraph.cpp:5-53.
// AlignRotations minimises sum_i Cauchy(0.1)(|| gt_i - Log(R_i * Exp(a)) ||^2) over the 3-vector a with Ceres' Levenberg-
// Marquardt defaults (500 iterations, function_tolerance 0); the functor is differentiated with forward-mode duals the way
// ceres::AutoDiffCostFunction does, so the zero-angle branches of ceres/rotation.h carry the same derivatives.
#include <algorithm>
#include <cmath>
#include <fstream>
#include <limits>
#include <sstream>

#include "../../include/gsfm/evaluation.hpp"

namespace gsfm {
namespace {

struct D3 {  // value + gradient w.r.t. the three alignment parameters
  double v, d[3];
  D3() : v(0), d{0, 0, 0} {}
  D3(double x) : v(x), d{0, 0, 0} {}  // NOLINT: implicit on purpose, constants mix freely with duals
};
inline D3 operator+(const D3& a, const D3& b) { D3 r; r.v = a.v + b.v; for (int k = 0; k < 3; ++k) r.d[k] = a.d[k] + b.d[k]; return r; }
inline D3 operator-(const D3& a, const D3& b) { D3 r; r.v = a.v - b.v; for (int k = 0; k < 3; ++k) r.d[k] = a.d[k] - b.d[k]; return r; }
inline D3 operator-(const D3& a) { D3 r; r.v = -a.v; for (int k = 0; k < 3; ++k) r.d[k] = -a.d[k]; return r; }
inline D3 operator*(const D3& a, const D3& b) { D3 r; r.v = a.v * b.v; for (int k = 0; k < 3; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k]; return r; }
inline D3 operator/(const D3& a, const D3& b) {
  D3 r; const double inv = 1.0 / b.v; r.v = a.v * inv;
  for (int k = 0; k < 3; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) * inv;
  return r;
}
inline bool operator>(const D3& a, const D3& b) { return a.v > b.v; }
inline bool operator<(const D3& a, const D3& b) { return a.v < b.v; }
inline bool operator>=(const D3& a, const D3& b) { return a.v >= b.v; }
inline D3 sqrt(const D3& a) { D3 r; r.v = std::sqrt(a.v); const double k = 0.5 / r.v; for (int c = 0; c < 3; ++c) r.d[c] = a.d[c] * k; return r; }
inline D3 sin(const D3& a) { D3 r; r.v = std::sin(a.v); const double k = std::cos(a.v); for (int c = 0; c < 3; ++c) r.d[c] = a.d[c] * k; return r; }
inline D3 cos(const D3& a) { D3 r; r.v = std::cos(a.v); const double k = -std::sin(a.v); for (int c = 0; c < 3; ++c) r.d[c] = a.d[c] * k; return r; }
inline D3 atan2(const D3& y, const D3& x) {
  D3 r; r.v = std::atan2(y.v, x.v);
  const double n = 1.0 / (x.v * x.v + y.v * y.v);
  for (int c = 0; c < 3; ++c) r.d[c] = (x.v * y.d[c] - y.v * x.d[c]) * n;
  return r;
}
using std::atan2; using std::cos; using std::sin; using std::sqrt;

// ceres::AngleAxisToRotationMatrix (rotation.h 1.14), row-major R
template <class T> void AngleAxisToMatrix(const T* a, T* R) {
  const T theta2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = sqrt(theta2);
    const T wx = a[0] / theta, wy = a[1] / theta, wz = a[2] / theta;
    const T c = cos(theta), s = sin(theta), k = T(1.0) - c;
    R[0] = c + wx * wx * k;       R[1] = wx * wy * k - wz * s;  R[2] = wy * s + wx * wz * k;
    R[3] = wz * s + wx * wy * k;  R[4] = c + wy * wy * k;       R[5] = -wx * s + wy * wz * k;
    R[6] = -wy * s + wx * wz * k; R[7] = wx * s + wy * wz * k;  R[8] = c + wz * wz * k;
  } else {  // first-order Taylor
    R[0] = T(1.0); R[1] = -a[2];  R[2] = a[1];
    R[3] = a[2];   R[4] = T(1.0); R[5] = -a[0];
    R[6] = -a[1];  R[7] = a[0];   R[8] = T(1.0);
  }
}
// ceres::RotationMatrixToAngleAxis = RotationMatrixToQuaternion + QuaternionToAngleAxis, row-major R
template <class T> void MatrixToAngleAxis(const T* R, T* a) {
  T q[4];
  const T trace = R[0] + R[4] + R[8];
  if (trace >= T(0.0)) {
    T t = sqrt(trace + T(1.0));
    q[0] = T(0.5) * t;
    t = T(0.5) / t;
    q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    T t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + T(1.0));
    q[i + 1] = T(0.5) * t;
    t = T(0.5) / t;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  const T s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > T(0.0)) {
    const T s = sqrt(s2);
    const T two_theta = T(2.0) * ((q[0] < T(0.0)) ? atan2(-s, -q[0]) : atan2(s, q[0]));
    const T k = two_theta / s;
    a[0] = q[1] * k; a[1] = q[2] * k; a[2] = q[3] * k;
  } else {
    a[0] = q[1] * T(2.0); a[1] = q[2] * T(2.0); a[2] = q[3] * T(2.0);
  }
}
template <class T> void Mul3(const T* A, const T* B, T* C) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// RotationAlignmentError (:22-69): residual = gt - Log(U * Exp(a))
struct Block { double U[9]; double gt[3]; };
void EvalBlock(const Block& b, const double* x, double* r, double* J /* 3x3 row-major, may be null */) {
  if (!J) {
    double A[9], M[9], aa[3];
    AngleAxisToMatrix(x, A);
    Mul3(b.U, A, M);
    MatrixToAngleAxis(M, aa);
    for (int k = 0; k < 3; ++k) r[k] = b.gt[k] - aa[k];
    return;
  }
  D3 xd[3], A[9], U[9], M[9], aa[3];
  for (int k = 0; k < 3; ++k) { xd[k] = D3(x[k]); xd[k].d[k] = 1.0; }
  for (int k = 0; k < 9; ++k) U[k] = D3(b.U[k]);
  AngleAxisToMatrix(xd, A);
  Mul3(U, A, M);
  MatrixToAngleAxis(M, aa);
  for (int k = 0; k < 3; ++k) { r[k] = b.gt[k] - aa[k].v; for (int c = 0; c < 3; ++c) J[3 * k + c] = -aa[k].d[c]; }
}

// cost = 1/2 sum rho(|r_i|^2), rho = ceres::CauchyLoss(0.1); fills the Corrector-scaled residuals and Jacobians if asked
double Evaluate(const std::vector<Block>& blocks, const double* x, std::vector<double>* res, std::vector<double>* jac) {
  const double b = 0.1 * 0.1, c = 1.0 / b;
  double cost = 0.0;
  for (size_t i = 0; i < blocks.size(); ++i) {
    double r[3], J[9];
    EvalBlock(blocks[i], x, r, jac ? J : nullptr);
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    const double sum = 1.0 + s * c, inv = 1.0 / sum;
    const double rho0 = b * std::log(sum), rho1 = std::max(std::numeric_limits<double>::min(), inv);
    cost += 0.5 * rho0;
    if (!jac) continue;
    // rho'' = -c inv^2 < 0: the Corrector reduces to sqrt(rho') on both residuals and Jacobian (corrector.cc:113-118)
    const double w = std::sqrt(rho1);
    for (int k = 0; k < 3; ++k) (*res)[3 * i + k] = w * r[k];
    for (int k = 0; k < 9; ++k) (*jac)[9 * i + k] = w * J[k];
  }
  return cost;
}

bool Solve3(const double* A /* sym 3x3 */, const double* rhs, double* x) {  // Cholesky
  const double l00 = std::sqrt(A[0]);
  if (!(l00 > 0.0)) return false;
  const double l10 = A[3] / l00, l20 = A[6] / l00;
  const double d1 = A[4] - l10 * l10;
  if (!(d1 > 0.0)) return false;
  const double l11 = std::sqrt(d1), l21 = (A[7] - l20 * l10) / l11;
  const double d2 = A[8] - l20 * l20 - l21 * l21;
  if (!(d2 > 0.0)) return false;
  const double l22 = std::sqrt(d2);
  const double y0 = rhs[0] / l00, y1 = (rhs[1] - l10 * y0) / l11, y2 = (rhs[2] - l20 * y0 - l21 * y1) / l22;
  x[2] = y2 / l22; x[1] = (y1 - l21 * x[2]) / l11; x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
  return true;
}

}  // namespace

double AngularDifference(const Eigen::Vector3d& rotation1, const Eigen::Vector3d& rotation2) {
  double R1[9], R2[9], L[9], aa[3];
  AngleAxisToMatrix(rotation1.data(), R1);
  AngleAxisToMatrix(rotation2.data(), R2);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L[3 * r + c] = R1[r] * R2[c] + R1[3 + r] * R2[3 + c] + R1[6 + r] * R2[6 + c];  // R1^T R2
  MatrixToAngleAxis(L, aa);
  return std::sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);  // Eigen::AngleAxisd(R).angle(), in [0, pi]
}

AlignmentSummary AlignRotations(const std::vector<Eigen::Vector3d>& gt_rotation, std::vector<Eigen::Vector3d>* rotation) {
  AlignmentSummary out;
  const size_t n = gt_rotation.size();
  if (!rotation || rotation->size() != n) { out.message = "size mismatch"; return out; }
  std::vector<Block> blocks(n);
  for (size_t i = 0; i < n; ++i) {
    AngleAxisToMatrix((*rotation)[i].data(), blocks[i].U);
    for (int k = 0; k < 3; ++k) blocks[i].gt[k] = gt_rotation[i][k];
  }
  // Ceres 1.14 trust-region Levenberg-Marquardt with Solver::Options defaults except max_num_iterations = 500,
  // function_tolerance = 0 (:161-164); DENSE_QR on three columns = the damped normal equations solved exactly.
  double x[3] = {0, 0, 0};
  std::vector<double> res(3 * n), jac(9 * n);
  double cost = Evaluate(blocks, x, &res, &jac);
  out.initial_cost = cost;
  double scale[3] = {1, 1, 1};
  {
    double col[3] = {0, 0, 0};
    for (size_t i = 0; i < 3 * n; ++i) for (int c = 0; c < 3; ++c) col[c] += jac[3 * i + c] * jac[3 * i + c];
    for (int c = 0; c < 3; ++c) scale[c] = 1.0 / (1.0 + std::sqrt(col[c]));  // jacobi_scaling
  }
  double radius = 1e4, decrease_factor = 2.0;
  const double gtol = 1e-10, ptol = 1e-8, min_rel_decrease = 1e-3;
  auto normal = [&](double* JtJ, double* Jtr) {
    for (int k = 0; k < 9; ++k) JtJ[k] = 0.0;
    for (int k = 0; k < 3; ++k) Jtr[k] = 0.0;
    for (size_t i = 0; i < 3 * n; ++i) {
      const double* j = &jac[3 * i];
      const double js[3] = {j[0] * scale[0], j[1] * scale[1], j[2] * scale[2]};
      for (int a = 0; a < 3; ++a) { Jtr[a] += js[a] * res[i]; for (int b2 = 0; b2 < 3; ++b2) JtJ[3 * a + b2] += js[a] * js[b2]; }
    }
  };
  double JtJ[9], Jtr[3];
  normal(JtJ, Jtr);
  auto gmax = [&]() { double m = 0; for (int c = 0; c < 3; ++c) m = std::max(m, std::fabs(Jtr[c] / scale[c])); return m; };
  if (n == 0 || gmax() <= gtol) { out.converged = true; out.final_cost = cost; out.message = "gradient tolerance at start"; }
  else {
    for (out.iterations = 0; out.iterations < 500; ++out.iterations) {
      double A[9], rhs[3], ds[3];
      for (int k = 0; k < 9; ++k) A[k] = JtJ[k];
      for (int c = 0; c < 3; ++c) { A[4 * c] += std::min(std::max(JtJ[4 * c], 1e-6), 1e32) / radius; rhs[c] = -Jtr[c]; }
      bool ok = Solve3(A, rhs, ds);
      double delta[3], model_change = 0.0;
      if (ok) {
        for (int c = 0; c < 3; ++c) delta[c] = ds[c] * scale[c];
        // -(|J d|^2 / 2 + r^T J d)
        double q = 0.0, l = 0.0;
        for (int a = 0; a < 3; ++a) { l += ds[a] * Jtr[a]; for (int b2 = 0; b2 < 3; ++b2) q += ds[a] * JtJ[3 * a + b2] * ds[b2]; }
        model_change = -(0.5 * q + l);
        ok = model_change > 0.0;
      }
      if (!ok) {  // invalid step: shrink the region (trust_region_minimizer.cc)
        radius /= decrease_factor; decrease_factor *= 2.0;
        if (radius < 1e-32) { out.message = "trust region collapsed"; break; }
        continue;
      }
      const double xn[3] = {x[0] + delta[0], x[1] + delta[1], x[2] + delta[2]};
      const double step_norm = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
      const double x_norm = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      if (step_norm <= ptol * (x_norm + ptol)) { out.converged = true; out.message = "parameter tolerance"; break; }
      const double new_cost = Evaluate(blocks, xn, nullptr, nullptr);
      const double rel = (cost - new_cost) / model_change;
      if (rel > min_rel_decrease) {
        for (int c = 0; c < 3; ++c) x[c] = xn[c];
        const double change = cost - new_cost;
        cost = Evaluate(blocks, x, &res, &jac);
        normal(JtJ, Jtr);
        const double t = 2.0 * rel - 1.0;
        radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
        decrease_factor = 2.0;
        if (gmax() <= gtol) { out.converged = true; out.message = "gradient tolerance"; ++out.iterations; break; }
        if (std::fabs(change) <= 0.0 * cost) { out.converged = true; out.message = "function tolerance"; ++out.iterations; break; }
      } else {
        radius /= decrease_factor; decrease_factor *= 2.0;
        if (radius < 1e-32) { out.message = "trust region collapsed"; break; }
      }
    }
    out.final_cost = cost;
    if (out.message.empty()) out.message = "iteration limit";
  }
  out.alignment = Eigen::Vector3d(x[0], x[1], x[2]);
  // ApplyRotationTransformation (:120-145): R_i <- R_i * Exp(a)
  double A[9];
  AngleAxisToMatrix(x, A);
  for (size_t i = 0; i < n; ++i) {
    double M[9], aa[3];
    Mul3(blocks[i].U, A, M);
    MatrixToAngleAxis(M, aa);
    (*rotation)[i] = Eigen::Vector3d(aa[0], aa[1], aa[2]);
  }
  return out;
}

void ColmapViewGraph::read_poses(const std::string& path) {
  // COLMAP images.txt: four comment lines (the 4th carries "Number of images: N"), then per image
  //   IMAGE_ID QW QX QY QZ TX TY TZ CAMERA_ID NAME / one line of 2-D points  (read_colmap_posegraph.cpp:7-53)
  std::ifstream fin(path);
  if (!fin.is_open()) throw std::runtime_error("cannot read " + path);
  std::string line;
  for (int k = 0; k < 4; ++k) std::getline(fin, line);
  size_t pos = line.find("Number of images:");
  int num = 0;
  if (pos != std::string::npos) num = std::atoi(line.c_str() + pos + 17);
  for (int i = 0; i < num; ++i) {
    uint32_t image_id, camera_id;
    double qw, qx, qy, qz, tx, ty, tz;
    std::string name;
    if (!(fin >> image_id >> qw >> qx >> qy >> qz >> tx >> ty >> tz >> camera_id >> name)) break;
    std::getline(fin, line);
    std::getline(fin, line);
    const size_t slash = name.find('/');
    if (slash != std::string::npos) name = name.substr(slash + 1);
    image_names[image_id] = name;
    image_ids[name] = image_id;
    // ceres::QuaternionToAngleAxis on (w, x, y, z)
    double aa[3];
    const double s2 = qx * qx + qy * qy + qz * qz;
    if (s2 > 0.0) {
      const double s = std::sqrt(s2);
      const double k = 2.0 * ((qw < 0.0) ? std::atan2(-s, -qw) : std::atan2(s, qw)) / s;
      aa[0] = qx * k; aa[1] = qy * k; aa[2] = qz * k;
    } else { aa[0] = 2 * qx; aa[1] = 2 * qy; aa[2] = 2 * qz; }
    poses[image_id] = {tx, ty, tz, aa[0], aa[1], aa[2]};
  }
  num_view = num;
}

namespace {
theia::ViewId ViewIdFromName(const theia::Reconstruction& rec, const std::string& name) {
  for (const auto& kv : rec.view_names) if (kv.second == name) return kv.first;
  return theia::kInvalidViewId;
}
CompareInfo CompareRotations(std::vector<Eigen::Vector3d>& reference, std::vector<Eigen::Vector3d>& estimate) {
  CompareInfo result;
  AlignRotations(reference, &estimate);
  for (size_t i = 0; i < reference.size(); ++i) result.rotation_diff_when_align.push_back(AngularDifference(reference[i], estimate[i]));
  result.common_camera = (int)reference.size();
  return result;
}
}  // namespace

std::vector<std::string> FindCommonEstimatedViewsByName(const theia::Reconstruction& a, const theia::Reconstruction& b) {
  std::vector<std::string> out;
  for (const auto& kv : a.view_names) {
    if (!a.orientation.count(kv.first)) continue;
    const theia::ViewId other = ViewIdFromName(b, kv.second);
    if (other != theia::kInvalidViewId && b.orientation.count(other)) out.push_back(kv.second);
  }
  std::sort(out.begin(), out.end());
  return out;
}
std::vector<std::string> FindCommonEstimatedViewsByNameColmap(const ColmapViewGraph& colmap, const theia::Reconstruction& rec) {
  std::vector<std::string> out;
  for (const auto& kv : colmap.image_names) {
    const theia::ViewId v = ViewIdFromName(rec, kv.second);
    if (v != theia::kInvalidViewId && rec.orientation.count(v)) out.push_back(kv.second);
  }
  std::sort(out.begin(), out.end());
  return out;
}
CompareInfo compare_orientations(const std::vector<std::string>& names, const theia::Reconstruction& reference, theia::Reconstruction* to_align, double) {
  std::vector<Eigen::Vector3d> r1, r2;
  for (const std::string& nm : names) {
    r1.push_back(reference.orientation.at(ViewIdFromName(reference, nm)));
    r2.push_back(to_align->orientation.at(ViewIdFromName(*to_align, nm)));
  }
  return CompareRotations(r1, r2);
}
CompareInfo compare_orientations_colmap(const std::vector<std::string>& names, const ColmapViewGraph& reference, theia::Reconstruction* to_align, double) {
  std::vector<Eigen::Vector3d> r1, r2;
  for (const std::string& nm : names) {
    const std::vector<double>& pose = reference.poses.at(reference.image_ids.at(nm));
    r1.emplace_back(pose[3], pose[4], pose[5]);
    r2.push_back(to_align->orientation.at(ViewIdFromName(*to_align, nm)));
  }
  return CompareRotations(r1, r2);
}

}  // namespace gsfm
