// pybind11 module `GlobalSfMpy`: the Python surface of the reference (bind_src/GlobalSfMpy.cpp:139-691)
// for the rotation-averaging path — everything scripts/sfm_pipeline.py touches up to its
// rotation-only exit (:31-70), the estimator classes, the loss trampoline and the gamma constants —
// backed by the MI355X solver.  Stages that are out of scope (tracks, BA, triangulation, image I/O)
// are not bound; calling them raises AttributeError instead of silently doing nothing.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <pybind11/stl_bind.h>

#include <cmath>
#include <cstring>
#include <fstream>
#include <memory>

#include "../../include/gsfm/GSfM_nonlinear_rotation_estimator.hpp"
#include "../../include/gsfm/evaluation.hpp"
#include "../../include/gsfm/view_graph.hpp"

namespace py = pybind11;
using namespace theia;

// numpy <-> the compat 3-vector / 3x3 (the reference relies on pybind11/eigen.h, bind :4)
namespace pybind11 { namespace detail {
template <> struct type_caster<Eigen::Vector3d> {
  PYBIND11_TYPE_CASTER(Eigen::Vector3d, _("numpy.ndarray[float64[3]]"));
  bool load(handle src, bool) {
    auto a = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(src);
    if (!a || a.size() != 3) return false;
    for (int k = 0; k < 3; ++k) value[k] = a.data()[k];
    return true;
  }
  static handle cast(const Eigen::Vector3d& v, return_value_policy, handle) {
    py::array_t<double> a(3);
    for (int k = 0; k < 3; ++k) a.mutable_data()[k] = v[k];
    return a.release();
  }
};
template <> struct type_caster<Eigen::Matrix3d> {
  PYBIND11_TYPE_CASTER(Eigen::Matrix3d, _("numpy.ndarray[float64[3,3]]"));
  bool load(handle src, bool) {
    auto a = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(src);
    if (!a || a.ndim() != 2 || a.shape(0) != 3 || a.shape(1) != 3) return false;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) value(r, c) = a.at(r, c);
    return true;
  }
  static handle cast(const Eigen::Matrix3d& m, return_value_policy, handle) {
    py::array_t<double> a({3, 3});
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a.mutable_at(r, c) = m(r, c);
    return a.release();
  }
};
}}  // namespace pybind11::detail

typedef std::unordered_map<uint32_t, Eigen::Vector3d> OrientationMap;
typedef std::unordered_map<ViewIdPair, TwoViewInfo> EdgeMap;
PYBIND11_MAKE_OPAQUE(OrientationMap);
PYBIND11_MAKE_OPAQUE(EdgeMap);
PYBIND11_MAKE_OPAQUE(CovarianceMap);
PYBIND11_MAKE_OPAQUE(std::vector<double>);

namespace {

// ceres::LossFunction trampoline (bind :33-65) that can also describe itself to the device through an
// optional Python method native_program() -> list of (kind, p0, p1, p2) (globalsfmpy_amd/loss_functions.py).
class PyLoss : public ceres::LossFunction, public gsfm::DescribedLoss {
 public:
  void Evaluate(double sq_norm, double out[3]) const override {
    py::gil_scoped_acquire gil;
    py::function f = py::get_override(static_cast<const ceres::LossFunction*>(this), "Evaluate");
    if (!f) py::pybind11_fail("Tried to call pure virtual function \"LossFunction::Evaluate\"");
    py::array_t<double> buf(3);
    buf.mutable_data()[0] = buf.mutable_data()[1] = buf.mutable_data()[2] = 0.0;
    f(sq_norm, buf);
    for (int k = 0; k < 3; ++k) out[k] = buf.data()[k];
  }
  int NativeProgram(gsfm_loss_node* out, int cap) const override {
    py::gil_scoped_acquire gil;
    py::function f = py::get_override(static_cast<const ceres::LossFunction*>(this), "native_program");
    if (!f) return -1;
    py::object r = f();
    if (r.is_none()) return -1;
    py::list nodes = r.cast<py::list>();
    if ((int)nodes.size() > cap) return -1;
    int n = 0;
    for (auto h : nodes) {
      py::tuple t = h.cast<py::tuple>();
      gsfm_loss_node nd{};
      nd.kind = t[0].cast<int>();
      for (size_t c = 1; c < t.size() && c < 4; ++c) nd.p[c - 1] = t[c].cast<double>();
      out[n++] = nd;
    }
    return n;
  }
};

class PyRotationEstimator : public RotationEstimator {
 public:
  bool EstimateRotations(const EdgeMap& view_pairs, OrientationMap* rotations) override {
    py::gil_scoped_acquire gil;
    PYBIND11_OVERRIDE_PURE(bool, RotationEstimator, EstimateRotations, view_pairs, rotations);
  }
};

// --- light stand-ins for the pipeline objects sfm_pipeline.py passes around ---
struct ReconstructionEstimatorOptions {
  int num_threads = 1;                                       // reconstruction_estimator_options.h:99
  int min_num_two_view_inliers = 30;                         // :107
  double rotation_filtering_max_difference_degrees = 5.0;    // :122
};
struct ReconstructionBuilderOptions {
  int num_threads = 1;
  ReconstructionEstimatorOptions reconstruction_estimator_options;
};
struct FeaturesAndMatchesDatabase {  // bind :166-170 wraps a RocksDB store; only its path matters to the rotation stage
  std::string path;
};
struct ReconstructionBuilder {
  ReconstructionBuilderOptions options;
  Reconstruction* reconstruction;
  ViewGraph* view_graph;
  // the (options, database) form owns what the COLMAP ingestion fills
  std::shared_ptr<Reconstruction> owned_reconstruction;
  std::shared_ptr<ViewGraph> owned_view_graph;
};

// The flattened inputs of gsfm_cov_estimate as numpy arrays, for inspection and for globalsfmpy_amd.covariance.
py::dict edge_matches_dict(const gsfm::EdgeMatches& em) {
  const py::ssize_t E = (py::ssize_t)em.edges.size();
  py::array_t<uint32_t> edges({E, (py::ssize_t)2});
  for (py::ssize_t e = 0; e < E; ++e) { edges.mutable_at(e, 0) = em.edges[e].first; edges.mutable_at(e, 1) = em.edges[e].second; }
  auto vec = [](const std::vector<double>& v, py::ssize_t cols) {
    py::array_t<double> a({(py::ssize_t)(v.size() / cols), cols});
    if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(double));
    return a;
  };
  py::array_t<uint64_t> ptr((py::ssize_t)em.match_ptr.size());
  std::memcpy(ptr.mutable_data(), em.match_ptr.data(), em.match_ptr.size() * sizeof(uint64_t));
  py::dict d;
  d["edges"] = edges; d["match_ptr"] = ptr; d["matches"] = vec(em.matches, 4); d["intrinsics"] = vec(em.intrinsics, 6);
  d["rot"] = vec(em.rotation, 3); d["trans"] = vec(em.position, 3);
  return d;
}

py::dict summary_dict(const gsfm_rot_summary& s) {
  py::dict d;
  d["termination"] = s.termination; d["num_iterations"] = s.num_iterations;
  d["num_successful_steps"] = s.num_successful_steps; d["num_unsuccessful_steps"] = s.num_unsuccessful_steps;
  d["num_residual_sweeps"] = s.num_residual_sweeps; d["num_cg_iterations"] = s.num_cg_iterations;
  d["iters_to_1e6"] = s.iters_to_1e6; d["outer_iterations"] = s.outer_iterations;
  d["initial_cost"] = s.initial_cost; d["final_cost"] = s.final_cost; d["t_total_ms"] = s.t_total_ms;
  d["num_edges_used"] = s.num_edges_used; d["last_weight_change"] = s.last_weight_change;
  d["num_dense_solves"] = s.num_dense_solves; d["num_linearizations"] = s.num_linearizations;
  d["t_linearize_ms"] = s.t_linearize_ms; d["t_sweep_ms"] = s.t_sweep_ms; d["t_cg_ms"] = s.t_cg_ms;
  d["num_inexact_steps"] = s.num_inexact_steps; d["num_forcing_refinements"] = s.num_forcing_refinements; d["num_forcing_restarts"] = s.num_forcing_restarts;
  d["num_pcg_capped_steps"] = s.num_pcg_capped_steps; d["worst_accepted_cg_residual"] = s.worst_accepted_cg_residual;
  return d;
}

// src/GSfM_global_reconstruction_estimator.cpp, the rotation stage only.
class GlobalReconstructionEstimator {
 public:
  explicit GlobalReconstructionEstimator(const ReconstructionEstimatorOptions& o) : options_(o) {}
  // Estimate_BeforeStep3 (:272-293 -> :369-395): min-inlier filter, largest connected component.
  // Camera calibration from priors is out of scope (no cameras here).
  bool FilterInitialViewGraphAndCalibrateCameras(ViewGraph* vg, Reconstruction* rec) {
    view_graph_ = vg; reconstruction_ = rec;
    bool any_counts = false;
    for (const auto& e : vg->GetAllEdges()) any_counts |= e.second.num_verified_matches > 0;
    if (any_counts) {  // match counts need tracks.txt; without them every edge would be dropped
      std::vector<ViewIdPair> drop;
      for (const auto& e : vg->GetAllEdges()) if (e.second.num_verified_matches < options_.min_num_two_view_inliers) drop.push_back(e.first);
      for (const auto& k : drop) vg->RemoveEdge(k.first, k.second);
    }
    RemoveDisconnectedViewPairs(vg);
    return vg->NumEdges() > 0;
  }
  bool InitOrientations() { return OrientationsFromMaximumSpanningTree(*view_graph_, &orientations_); }
  bool EstimateGlobalRotationsNonLinear(ceres::LossFunction* loss, RotationErrorType type) {  // :440-461
    if (!view_graph_) throw std::runtime_error("call FilterInitialViewGraphAndCalibrateCameras first");
    InitOrientations();
    GSfMNonlinearRotationEstimator est;
    const bool ok = est.EstimateRotationsWithCustomizedLoss(view_graph_->GetAllEdges(), &orientations_, loss, options_.num_threads, type);
    summary_ = est.LastSummary(); error_ = est.LastError();
    return ok;
  }
  bool EstimateGlobalRotationsUncertainty(ceres::LossFunction* loss, CovarianceMap& cov, RotationErrorType type) {  // :463-485
    if (!view_graph_) throw std::runtime_error("call FilterInitialViewGraphAndCalibrateCameras first");
    InitOrientations();
    GSfMNonlinearRotationEstimator est;
    const bool ok = est.EstimateRotationsWithCustomizedLossAndCovariance(view_graph_->GetAllEdges(), &orientations_, loss,
                                                                          options_.num_threads, cov, type, nullptr);
    summary_ = est.LastSummary(); error_ = est.LastError();
    return ok;
  }
  bool EstimateGlobalRotationsSigmaConsensus(ceres::LossFunction* loss, int iters, double sigma_max) {  // :487-507
    if (!view_graph_) throw std::runtime_error("call FilterInitialViewGraphAndCalibrateCameras first");
    InitOrientations();
    GSfMNonlinearRotationEstimator est;
    const bool ok = est.EstimateRotationsWithSigmaConsensus(view_graph_->GetAllEdges(), &orientations_, loss, options_.num_threads, iters, sigma_max);
    summary_ = est.LastSummary(); error_ = est.LastError();
    return ok;
  }
  void FilterRotations() {  // :509-524
    FilterViewPairsFromOrientation(orientations_, options_.rotation_filtering_max_difference_degrees, view_graph_);
    for (ViewId v : RemoveDisconnectedViewPairs(view_graph_)) orientations_.erase(v);
  }
  ReconstructionEstimatorOptions options_;
  ViewGraph* view_graph_ = nullptr;
  Reconstruction* reconstruction_ = nullptr;
  OrientationMap orientations_;
  gsfm_rot_summary summary_{};
  std::string error_;
};

void load_1dsfm_config(const std::string& flagfile, ReconstructionBuilderOptions& options) {
  // bind :185-271 parses the YAML with yaml-cpp; only the keys the rotation stage consumes are read here.
  py::gil_scoped_acquire gil;
  py::object cfg = py::module_::import("yaml").attr("safe_load")(py::module_::import("builtins").attr("open")(flagfile));
  auto get = [&](const char* k, auto& dst) { if (cfg.contains(k) && !cfg[k].is_none()) dst = cfg[k].cast<std::decay_t<decltype(dst)>>(); };
  get("num_threads", options.num_threads);
  options.reconstruction_estimator_options.num_threads = options.num_threads;
  get("min_num_inliers_for_valid_match", options.reconstruction_estimator_options.min_num_two_view_inliers);
  get("post_rotation_filtering_degrees", options.reconstruction_estimator_options.rotation_filtering_max_difference_degrees);
}

}  // namespace

PYBIND11_MODULE(_GlobalSfMpy, m) {  // imported through the GlobalSfMpy.py shim beside it (HIP runtime pre-load)
  m.doc() = "MI355X-native drop-in for the rotation-averaging surface of zhangganlin/GlobalSfMpy";

  py::bind_map<OrientationMap>(m, "MapViewIdVector3d");
  py::bind_map<EdgeMap>(m, "MapEdges");
  py::bind_map<CovarianceMap>(m, "MapEdgesCovariance");
  py::bind_vector<std::vector<double>>(m, "VectorDouble");

  py::enum_<RotationErrorType>(m, "RotationErrorType", py::arithmetic())  // the 7 values the reference binds (:431-439)
      .value("QUATERNION_COSINE", RotationErrorType::QUATERNION_COSINE)
      .value("QUATERNION_NORM", RotationErrorType::QUATERNION_NORM)
      .value("ROTATION_MAT_FNORM", RotationErrorType::ROTATION_MAT_FNORM)
      .value("ANGLE_AXIS_COVARIANCE", RotationErrorType::ANGLE_AXIS_COVARIANCE)
      .value("ANGLE_AXIS", RotationErrorType::ANGLE_AXIS)
      .value("ANGLE_AXIS_COVTRACE", RotationErrorType::ANGLE_AXIS_COVTRACE)
      .value("ANGLE_AXIS_COVNORM", RotationErrorType::ANGLE_AXIS_COVNORM);
  enum class PositionErrorType { BASELINE = 0 };
  py::enum_<PositionErrorType>(m, "PositionErrorType", py::arithmetic()).value("BASELINE", PositionErrorType::BASELINE);

  py::class_<ceres::LossFunction, PyLoss>(m, "LossFunction").def(py::init<>());

  py::class_<RotationEstimator, PyRotationEstimator>(m, "RotationEstimator")
      .def(py::init<>())
      .def("EstimateRotations", &RotationEstimator::EstimateRotations);

  py::class_<GSfMNonlinearRotationEstimator, RotationEstimator>(m, "NonlinearRotationEstimator")
      .def(py::init<>())
      .def(py::init<const double>())
      .def("EstimateRotations", &GSfMNonlinearRotationEstimator::EstimateRotations, py::call_guard<py::gil_scoped_release>())
      .def("EstimateRotationsWithCustomizedLoss", &GSfMNonlinearRotationEstimator::EstimateRotationsWithCustomizedLoss,
           py::arg("view_pairs"), py::arg("global_orientations"), py::arg("loss_function"), py::arg("thread_num"),
           py::arg("rotation_error_type") = RotationErrorType::QUATERNION_COSINE, py::call_guard<py::gil_scoped_release>())
      .def("EstimateRotationsWithCustomizedLossAndCovariance",
           [](GSfMNonlinearRotationEstimator& e, const EdgeMap& vp, OrientationMap* o, ceres::LossFunction* l, int threads, CovarianceMap& c,
              RotationErrorType t) { py::gil_scoped_release rel; return e.EstimateRotationsWithCustomizedLossAndCovariance(vp, o, l, threads, c, t, nullptr); })
      .def("EstimateRotationsWithSigmaConsensus", &GSfMNonlinearRotationEstimator::EstimateRotationsWithSigmaConsensus,
           py::call_guard<py::gil_scoped_release>())
      .def("LastSummary", [](const GSfMNonlinearRotationEstimator& e) { return summary_dict(e.LastSummary()); })
      .def("LastError", [](const GSfMNonlinearRotationEstimator& e) { return std::string(e.LastError()); });

  m.def("test_loss_with_input_x", [](ceres::LossFunction* loss, double x) {  // bind :179-183
    double out[3] = {0, 0, 0};
    loss->Evaluate(x, out);
    py::print("[" + std::to_string(out[0]) + ", " + std::to_string(out[1]) + ", " + std::to_string(out[2]) + "]");
  });

  py::class_<ReconstructionEstimatorOptions>(m, "ReconstructionEstimatorOptions")
      .def(py::init<>())
      .def_readwrite("num_threads", &ReconstructionEstimatorOptions::num_threads)
      .def_readwrite("min_num_two_view_inliers", &ReconstructionEstimatorOptions::min_num_two_view_inliers)
      .def_readwrite("rotation_filtering_max_difference_degrees", &ReconstructionEstimatorOptions::rotation_filtering_max_difference_degrees);
  py::class_<ReconstructionBuilderOptions>(m, "ReconstructionBuilderOptions")
      .def(py::init<>())
      .def_readwrite("num_threads", &ReconstructionBuilderOptions::num_threads)
      .def_readwrite("reconstruction_estimator_options", &ReconstructionBuilderOptions::reconstruction_estimator_options);
  m.def("load_1DSFM_config", &load_1dsfm_config);

  py::class_<Reconstruction>(m, "Reconstruction")
      .def(py::init<>())
      .def("NumTracks", &Reconstruction::NumTracks)
      .def("NumViews", &Reconstruction::NumViews)
      .def("EstimatedOrientations", [](const Reconstruction& r) { return r.orientation; })
      .def("ViewNames", [](const Reconstruction& r) { return r.view_names; })
      .def("SetViewName", [](Reconstruction& r, ViewId v, const std::string& name) { r.views.insert(v); r.view_names[v] = name; })
      .def("SetOrientation", [](Reconstruction& r, ViewId v, const Eigen::Vector3d& aa) { r.views.insert(v); r.orientation[v] = aa; })
      .def("MatchedFeatures", [](const Reconstruction& r) { return r.matches ? py::object(edge_matches_dict(*r.matches)) : py::object(py::none()); })
      .def("NumMatchedPairs", [](const Reconstruction& r) { return r.matches ? (int)r.matches->edges.size() : 0; });

  py::class_<ViewGraph>(m, "ViewGraph")
      .def(py::init<>())
      .def("NumViews", &ViewGraph::NumViews)
      .def("NumEdges", &ViewGraph::NumEdges)
      .def("HasView", &ViewGraph::HasView)
      .def("HasEdge", &ViewGraph::HasEdge)
      .def("ViewIds", &ViewGraph::ViewIds)
      .def("AddEdge", &ViewGraph::AddEdge)
      .def("RemoveEdge", &ViewGraph::RemoveEdge)
      .def("GetEdge", &ViewGraph::GetEdge, py::return_value_policy::reference_internal)
      .def("GetAllEdges", &ViewGraph::GetAllEdges, py::return_value_policy::reference);

  py::class_<TwoViewInfo>(m, "TwoViewInfo")
      .def(py::init<>())
      .def_readwrite("focal_length_1", &TwoViewInfo::focal_length_1)
      .def_readwrite("focal_length_2", &TwoViewInfo::focal_length_2)
      .def_readwrite("position_2", &TwoViewInfo::position_2)
      .def_readwrite("rotation_2", &TwoViewInfo::rotation_2)
      .def_readwrite("num_verified_matches", &TwoViewInfo::num_verified_matches)
      .def_readwrite("num_homography_inliers", &TwoViewInfo::num_homography_inliers)
      .def_readwrite("visibility_score", &TwoViewInfo::visibility_score);

  py::class_<ReconstructionBuilder>(m, "ReconstructionBuilder")
      .def(py::init([](ReconstructionBuilderOptions& o, Reconstruction* r, ViewGraph* g) { return new ReconstructionBuilder{o, r, g}; }),
           py::keep_alive<1, 3>(), py::keep_alive<1, 4>())
      .def(py::init([](ReconstructionBuilderOptions& o, FeaturesAndMatchesDatabase*) {  // sfm_pipeline.py:46
        auto* b = new ReconstructionBuilder{o, nullptr, nullptr, std::make_shared<Reconstruction>(), std::make_shared<ViewGraph>()};
        b->reconstruction = b->owned_reconstruction.get();
        b->view_graph = b->owned_view_graph.get();
        return b;
      }))
      .def("CheckView", [](ReconstructionBuilder& b) {  // views of the graph become views of the reconstruction
        for (ViewId v : b.view_graph->ViewIds()) b.reconstruction->views.insert(v);
      })
      // reference_internal: the (options, database) form owns both objects, so they must keep the builder alive
      .def("get_view_graph", [](ReconstructionBuilder& b) { return b.view_graph; }, py::return_value_policy::reference_internal)
      .def("get_reconstruction", [](ReconstructionBuilder& b) { return b.reconstruction; }, py::return_value_policy::reference_internal);

  py::class_<GlobalReconstructionEstimator>(m, "GlobalReconstructionEstimator")
      .def(py::init<const ReconstructionEstimatorOptions&>())
      .def("get_view_graph", [](GlobalReconstructionEstimator* e) { return e->view_graph_; }, py::return_value_policy::reference)
      .def("get_reconstruction", [](GlobalReconstructionEstimator* e) { return e->reconstruction_; }, py::return_value_policy::reference)
      .def_readwrite("orientations", &GlobalReconstructionEstimator::orientations_)
      .def_readwrite("options", &GlobalReconstructionEstimator::options_)
      .def("FilterInitialViewGraphAndCalibrateCameras", &GlobalReconstructionEstimator::FilterInitialViewGraphAndCalibrateCameras,
           py::keep_alive<1, 2>(), py::keep_alive<1, 3>())
      .def("OrientationsFromMaximumSpanningTree", &GlobalReconstructionEstimator::InitOrientations)
      .def("EstimateGlobalRotations", &GlobalReconstructionEstimator::EstimateGlobalRotationsNonLinear, py::arg("loss_func") = nullptr,
           py::arg("rotation_error_type") = RotationErrorType::QUATERNION_COSINE, py::call_guard<py::gil_scoped_release>())
      .def("EstimateGlobalRotationsUncertainty", &GlobalReconstructionEstimator::EstimateGlobalRotationsUncertainty,
           py::call_guard<py::gil_scoped_release>())
      .def("EstimateGlobalRotationsWithSigmaConsensus", &GlobalReconstructionEstimator::EstimateGlobalRotationsSigmaConsensus,
           py::call_guard<py::gil_scoped_release>())
      .def("FilterRotations", &GlobalReconstructionEstimator::FilterRotations)
      .def("LastSummary", [](const GlobalReconstructionEstimator& e) { return summary_dict(e.summary_); })
      .def("LastError", [](const GlobalReconstructionEstimator& e) { return e.error_; });

  m.def("Read1DSFM", [](const std::string& dir, Reconstruction* rec, ViewGraph* vg, CovarianceMap& cov) {  // bind :611-617
    std::string err;
    if (!gsfm::Read1DSFMViewGraph(dir, vg, &err)) throw std::runtime_error(err);
    for (ViewId v : vg->ViewIds()) rec->views.insert(v);
    gsfm::Tracks1DSfM tracks;
    if (gsfm::Read1DSFMTracks(dir, &tracks, nullptr)) {  // keypoints + tracks, when the dataset has them
      auto em = std::make_shared<gsfm::EdgeMatches>();
      gsfm::CollectEdgeMatches(tracks, *vg, em.get());
      rec->matches = em;
      for (const auto& kv : tracks.names) if (rec->views.count(kv.first)) rec->view_names[kv.first] = kv.second;
    }
    for (ViewId v : rec->views) if (!rec->view_names.count(v)) rec->view_names[v] = std::to_string(v);  // no list.txt: the id is the name
    gsfm::ReadCovariance(dir, &cov);
  });
  py::class_<FeaturesAndMatchesDatabase>(m, "FeaturesAndMatchesDatabase").def(py::init([](const std::string& p) { return new FeaturesAndMatchesDatabase{p}; }));
  // src/read_colmap_posegraph.cpp:55-164
  m.def("AddColmapMatchesToReconstructionBuilder", [](const std::string& two_views, const std::string& images_wildcard, ReconstructionBuilder* b) {
    gsfm::ColmapPoseGraph g;
    std::string err;
    if (!gsfm::ReadColmapTwoViews(two_views, gsfm::ExpandWildcard(images_wildcard), &g, &err)) throw std::runtime_error(err);
    for (const auto& e : g.view_graph.GetAllEdges()) b->view_graph->AddEdge(e.first.first, e.first.second, e.second);
    for (size_t v = 0; v < g.view_names.size(); ++v) { b->reconstruction->views.insert((ViewId)v); b->reconstruction->view_names[(ViewId)v] = g.view_names[v]; }
    b->reconstruction->matches = std::make_shared<gsfm::EdgeMatches>(std::move(g.matches));
  });
  // bind :286-287 -> src/uncertainty.cpp:164-198; returns the counters (the reference returns None)
  m.def("store_covariance_rot", [](const std::string& dir, Reconstruction* rec, ViewGraph* vg) {
    if (!rec->matches) throw std::runtime_error("store_covariance_rot: the reconstruction carries no matched features (dataset without tracks)");
    gsfm::CalcCovarianceStats st;
    std::string err;
    {
      py::gil_scoped_release release;
      if (!gsfm::StoreCovarianceRot(dir, *rec->matches, *vg, nullptr, &st, &err)) { py::gil_scoped_acquire a; throw std::runtime_error(err); }
    }
    py::dict d;
    d["num_edges"] = st.num_edges; d["num_matches"] = st.num_matches; d["num_written"] = st.num_written;
    d["num_skipped"] = st.num_skipped; d["num_singular"] = st.num_singular; d["kernel_ms"] = st.kernel_ms;
    return d;
  });
  m.def("ReadCovariance", [](const std::string& dir, CovarianceMap& cov) { gsfm::ReadCovariance(dir, &cov); });
  m.def("WriteCovariance", [](const std::string& dir, const CovarianceMap& cov) { return gsfm::WriteCovariance(dir, cov); });
  // The flattened inputs CalcCovariance hands to gsfm_cov_estimate, for inspection and for globalsfmpy_amd.covariance.
  m.def("Read1DSFMEdgeMatches", [](const std::string& dir) {
    gsfm::Tracks1DSfM tracks;
    theia::ViewGraph vg;
    std::string err;
    if (!gsfm::Read1DSFMTracks(dir, &tracks, &err) || !gsfm::Read1DSFMViewGraph(dir, &vg, &err)) throw std::runtime_error(err);
    gsfm::EdgeMatches em;
    gsfm::CollectEdgeMatches(tracks, vg, &em);
    return edge_matches_dict(em);
  });
  // bind :623-628: estimates every edge's rotation covariance on the device and writes <dir>/covariance_rot.txt;
  // returns the counters (the reference returns None).
  m.def("CalcCovariance", [](const std::string& dir) {
    gsfm::CalcCovarianceStats st;
    std::string err;
    {
      py::gil_scoped_release release;
      if (!gsfm::CalcCovariance(dir, nullptr, &st, &err)) { py::gil_scoped_acquire a; throw std::runtime_error(err); }
    }
    py::dict d;
    d["num_edges"] = st.num_edges; d["num_matches"] = st.num_matches; d["num_written"] = st.num_written;
    d["num_skipped"] = st.num_skipped; d["num_singular"] = st.num_singular; d["kernel_ms"] = st.kernel_ms;
    return d;
  });
  m.def("OrientationsFromMaximumSpanningTree", [](const ViewGraph& vg, OrientationMap* o) { return OrientationsFromMaximumSpanningTree(vg, o); });
  m.def("FilterViewPairsFromOrientation", &FilterViewPairsFromOrientation);
  // bind :658, src/compare_reconstructions.cpp:617-647: (view_graph, reconstruction_to_eval, covariances, residuals) -- fills `residuals`
  m.def("residuals_of_relative_rot", [](const ViewGraph& vg, const Reconstruction& rec, const CovarianceMap& cov, std::vector<double>& residuals) {
    gsfm::ResidualsOfRelativeRotations(vg, rec.orientation, cov, &residuals);
  }, py::call_guard<py::gil_scoped_release>());
  m.def("SetOrientations", [](const OrientationMap& o, Reconstruction* rec) {  // bind :80-98
    rec->orientation.clear();
    for (const auto& kv : o) if (rec->views.count(kv.first)) rec->orientation[kv.first] = kv.second;
  });
  m.def("InitGlog", [](int, bool, std::string) {}, py::arg("log_level") = 0, py::arg("logtostderr") = true, py::arg("log_dir") = "./log");
  m.def("StopGlog", []() {});
  m.def("WriteReconstruction", [](const Reconstruction& rec, const std::string& path) {
    std::ofstream f(path);
    f << "# view_id angle_axis(3)  -- rotation-only reconstruction written by the MI355X build\n";
    for (const auto& kv : rec.orientation) f << kv.first << " " << kv.second[0] << " " << kv.second[1] << " " << kv.second[2] << "\n";
    return (bool)f;
  });
  // bind :631 / Theia io/write_ply_file.cc:74-123: tracks (3-D points) and the positions of the estimated views as green vertices.
  // A rotation-only reconstruction has no tracks and no camera positions, so the file holds one green vertex at the origin per
  // estimated view -- the same header, the same vertex format, so sfm_pipeline.py's __main__ (:146) runs through.
  m.def("WritePlyFile", [](const std::string& ply_file, const Reconstruction& rec, int /*min_num_observations_per_point*/) {
    if (ply_file.empty()) throw std::invalid_argument("WritePlyFile: empty file name");
    std::ofstream f(ply_file);
    if (!f.is_open()) return false;
    size_t n = 0;
    for (const auto& kv : rec.orientation) n += rec.views.count(kv.first);
    f << "ply\nformat ascii 1.0\nelement vertex " << n
      << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header" << std::endl;
    for (const auto& kv : rec.orientation) if (rec.views.count(kv.first)) f << "0 0 0 0 255 0\n";
    return (bool)f;
  }, py::call_guard<py::gil_scoped_release>());
  // ---- evaluation (bind :387-394, :396-403, :651-664; orientations only) ----
  py::class_<gsfm::CompareInfo>(m, "CompareInfo")
      .def(py::init<>())
      .def_readwrite("rotation_diff_when_align", &gsfm::CompareInfo::rotation_diff_when_align)
      .def_readwrite("position_errors", &gsfm::CompareInfo::position_errors)
      .def_readwrite("num_3d_points", &gsfm::CompareInfo::num_3d_points)
      .def_readwrite("common_camera", &gsfm::CompareInfo::common_camera)
      .def_readwrite("num_reconstructed_view", &gsfm::CompareInfo::num_reconstructed_view);
  py::class_<gsfm::ColmapViewGraph>(m, "ColmapViewGraph")
      .def(py::init<>())
      .def("read_poses", &gsfm::ColmapViewGraph::read_poses)
      .def_readwrite("num_view", &gsfm::ColmapViewGraph::num_view)
      .def_readwrite("image_ids", &gsfm::ColmapViewGraph::image_ids)
      .def_readwrite("image_names", &gsfm::ColmapViewGraph::image_names)
      .def_readwrite("poses", &gsfm::ColmapViewGraph::poses);
  m.def("AngularDifference", &gsfm::AngularDifference);
  m.def("AlignRotations", [](const std::vector<Eigen::Vector3d>& gt, std::vector<Eigen::Vector3d> rot) {
    const gsfm::AlignmentSummary s = gsfm::AlignRotations(gt, &rot);
    py::dict d;
    d["alignment"] = s.alignment; d["initial_cost"] = s.initial_cost; d["final_cost"] = s.final_cost;
    d["iterations"] = s.iterations; d["converged"] = s.converged; d["message"] = s.message;
    return py::make_tuple(rot, d);
  }, "Returns (aligned rotations, summary); the reference aligns in place.");
  m.def("FindCommonViewsByName", &gsfm::FindCommonEstimatedViewsByName);
  m.def("FindCommonViewsByNameColmap", &gsfm::FindCommonEstimatedViewsByNameColmap);
  m.def("compare_orientations", &gsfm::compare_orientations);
  m.def("compare_orientations_colmap", &gsfm::compare_orientations_colmap);
  m.def("ReadImageSize", [](const std::string& path) {
    int w = 0, h = 0;
    if (!gsfm::ReadImageSize(path, &w, &h)) throw std::runtime_error("cannot read the size of image " + path);
    return py::make_tuple(w, h);
  });
  m.def("tgamma", [](double x) { return std::tgamma(x); });

  // gamma constants / tables (bind :663-686), regenerated by the library
  for (int nu : {3, 4, 9}) {
    const std::string s = std::to_string(nu);
    double C, q, gk;
    gsfm_magsac_constants(nu, &C, &q, &gk);
    const int n = gsfm_magsac_table(nu, nullptr, 0);
    std::vector<double> t(n);
    gsfm_magsac_table(nu, t.data(), n);
    m.attr(("nu" + s).c_str()) = py::float_((double)nu);
    m.attr(("stored_gamma_values" + s).c_str()) = py::cast(t);
    m.attr(("C" + s).c_str()) = py::float_(C);
    m.attr(("sigma_quantile" + s).c_str()) = py::float_(q);
    m.attr(("upper_incomplete_gamma_of_k" + s).c_str()) = py::float_(gk);
    m.attr(("stored_gamma_number" + s).c_str()) = py::int_(n);
    m.attr(("precision_of_stored_gamma" + s).c_str()) = py::float_(1000.0);
  }
}
