// C-ABI implementation (include/gsfm_rot.h): problem assembly, Levenberg-Marquardt control with
// Ceres 1.14 trust-region semantics, block-Jacobi PCG orchestration.  All arithmetic on the edges
// and cameras runs in the kernels of kernels.hpp; the host only sequences launches and reads a
// handful of scalars per LM iteration.  Built with hipcc --offload-arch=gfx950 into libgsfm_rot.so.
#include "host_common.hpp"
#include "solver_launch.hpp"
#include "solver_pcg.hpp"
#include "solver_dense.hpp"
#include "solver_components.hpp"
#include "solver_lm.hpp"
#include "problem_create.hpp"

// =============================================================================================
extern "C" {

int gsfm_rot_abi_version(void) { return GSFM_ROT_ABI_VERSION; }
const char* gsfm_last_error(void) { return g_err.c_str(); }

void gsfm_rot_options_default(gsfm_rot_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 200; o->num_threads = 1;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->jacobi_scaling = 1; o->max_cg_iterations = 20000; o->cg_relative_tolerance = 1e-12; o->cg_check_interval = 8; o->verbose = 0; o->pcg_single_reduction = -1; o->cg_stall_iterations = 0; o->dense_cholesky_max_cams = 512; o->pcg_hip_graph = 1;
  o->pcg_forcing = 1; o->pcg_forcing_tolerance = 1e-8; o->dense_cholesky_auto_cams = 5333; o->lm_device_control = 1; o->component_rest = 1;
}

int32_t gsfm_rot_residual_dim(int32_t t) { return t == GSFM_ROT_QUATERNION_NORM ? 4 : t == GSFM_ROT_ROTATION_MAT_FNORM ? 9 : 3; }

gsfm_status gsfm_rot_problem_create(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j, const double* rel_aa, int32_t error_type,
                                    const double* cov6, const double* inlier_weight, const gsfm_rot_shard* shard, gsfm_rot_problem** out) {
  // The host-side structure build allocates O(E) vectors and starts threads: an exception (std::bad_alloc, std::system_error) must not
  // cross the C boundary.  (On a sharded problem the peers of a rank that fails THIS way are not told: they wait in the agreement.)
  gsfm_rot_problem* live = nullptr;
  try {
    return problem_create_impl(n_cams, n_edges, edge_i, edge_j, rel_aa, error_type, cov6, inlier_weight, shard, out, &live);
  } catch (const std::exception& e) {
    if (live) gsfm_rot_problem_destroy(live);
    if (out) *out = nullptr;
    return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, std::string("problem creation ran out of host resources: ") + e.what());
  }
}

void gsfm_rot_problem_destroy(gsfm_rot_problem* P) {
  if (!P) return;
  DeviceGuard g(P->device);
  P->timer.destroy();
  P->pcg_graph.reset(); P->pcg2_graph.reset();
  if (P->dense_graph) (void)hipGraphExecDestroy(P->dense_graph);
  P->comps.drop_graph(); P->comps.drop_side();
  if (P->own_stream && P->stream) (void)hipStreamDestroy(P->stream);
  if (P->pin) (void)hipHostFree(P->pin);
  if (P->rec_host) (void)hipHostFree(P->rec_host);
  if (P->mail_host) (void)hipHostFree(P->mail_host);
  delete P;
}

gsfm_status gsfm_rot_set_stream(gsfm_rot_problem* P, void* s) {
  if (!P) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL problem");
  DeviceGuard g(P->device);
  if (P->own_stream && P->stream) { (void)hipStreamSynchronize(P->stream); (void)hipStreamDestroy(P->stream); }
  P->pcg_graph.reset(); P->pcg_graph.unusable = false; P->pcg2_graph.reset(); P->pcg2_graph.unusable = false;
  if (P->dense_graph) { (void)hipGraphExecDestroy(P->dense_graph); P->dense_graph = nullptr; }
  P->comps.drop_graph(); P->comps.drop_side();
  if (s) { P->stream = (hipStream_t)s; P->own_stream = false; }
  else { if (hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "hipStreamCreate failed"); P->own_stream = true; }
  P->timer.stream = P->stream;
  return GSFM_OK;
}

gsfm_status gsfm_rot_set_loss(gsfm_rot_problem* P, const gsfm_loss_node* prog, int32_t n) {
  if (!P || (n > 0 && !prog)) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  P->cb = nullptr;
  return (gsfm_status)prepare_loss(P, prog, n);
}

gsfm_status gsfm_rot_set_loss_callback(gsfm_rot_problem* P, gsfm_loss_callback fn, void* user) {
  if (!P || !fn) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  if (!P->rho_ext.p) {
    if (P->rho_ext.alloc(3 * P->n_edges_in) != hipSuccess || P->s_ext.alloc(P->n_edges_in) != hipSuccess)
      return (gsfm_status)fail(GSFM_ERR_HIP, "allocating callback-loss buffers failed");
  }
  P->cb = fn; P->cb_user = user;
  return GSFM_OK;
}

gsfm_status gsfm_rot_set_edge_weights(gsfm_rot_problem* P, const double* w) {
  if (!P || !w) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (P->functor != F_AA || P->wmode == W_MATRIX) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "edge weights only apply to the scalar-weight angle-axis types");
  DeviceGuard g(P->device);
  if (P->wmode == W_NONE) {  // ANGLE_AXIS: promote to a scalar-weight problem
    if (P->cost.ws.alloc(P->cost.n) != hipSuccess || P->dir.ws.alloc(P->dir.n) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc weight planes");
    P->wmode = W_SCALAR;
  }
  if (!P->w_orig.p && P->w_orig.alloc(P->n_edges_in) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc weights");
  if (hipMemcpyAsync(P->w_orig.p, w, 8 * P->n_edges_in, hipMemcpyHostToDevice, P->stream) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "upload weights");
  if (P->cost.n) hipLaunchKernelGGL(k_gather_weights, dim3(grid_for(P->cost.n)), dim3(GSFM_BLOCK), 0, P->stream, P->w_orig.p, P->cost.eid.p, P->cost.n, P->cost.ws.p);
  if (P->dir.n) hipLaunchKernelGGL(k_gather_weights, dim3(grid_for(P->dir.n)), dim3(GSFM_BLOCK), 0, P->stream, P->w_orig.p, P->dir.eid.p, P->dir.n, P->dir.ws.p);
  return (gsfm_status)sync_check(P, "set_edge_weights");
}

static gsfm_status solve_impl(gsfm_rot_problem* P, double* rot, const gsfm_rot_options* opt, gsfm_rot_summary* summary, bool resident);
gsfm_status gsfm_rot_solve(gsfm_rot_problem* P, double* rot, const gsfm_rot_options* opt, gsfm_rot_summary* summary) { return solve_impl(P, rot, opt, summary, false); }
// rot: DEVICE memory on the problem's device, caller numbering, 3 doubles per camera, in / out.  Returns with the problem's stream synchronised.
gsfm_status gsfm_rot_solve_resident(gsfm_rot_problem* P, double* rot_dev, const gsfm_rot_options* opt, gsfm_rot_summary* summary) {
  if (P && rot_dev) {
    DeviceGuard g(P->device);
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, rot_dev) != hipSuccess || (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged)) {
      (void)hipGetLastError();
      return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "gsfm_rot_solve_resident: rot_aa_dev_inout is not device memory (host arrays go to gsfm_rot_solve)");
    }
  }
  return solve_impl(P, rot_dev, opt, summary, true);
}
static gsfm_status solve_impl(gsfm_rot_problem* P, double* rot, const gsfm_rot_options* opt, gsfm_rot_summary* summary, bool resident) {
  if (!P || !rot) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  const gsfm_rot_options o = opt ? *opt : default_options();
  gsfm_rot_summary local; if (!summary) summary = &local;
  const double t0 = now_ms();
  if (int st = resident ? upload_state_resident(P, rot) : upload_state(P, rot)) return (gsfm_status)st;
  int st = lm_solve(P, o, summary);
  if (st == GSFM_INTERNAL_RESTART) {
    // The forcing schedule gave up on this trajectory after inexact steps had been applied (solver_lm.hpp, the contraction gate): the solve is
    // redone from the caller's rotations -- still untouched in `rot` -- with every step exact; what the abandoned attempt spent stays on the bill.
    const gsfm_rot_summary spent = *summary;
    gsfm_rot_options o2 = o;
    o2.pcg_forcing = 0;
    if (o.verbose) fprintf(stderr, "[gsfm] forcing schedule abandoned after %d LM iterations (steps stopped contracting): restarting with exact steps\n", spent.num_iterations);
    if (int st2 = resident ? upload_state_resident(P, rot) : upload_state(P, rot)) return (gsfm_status)st2;
    st = lm_solve(P, o2, summary);
    summary->num_forcing_restarts = 1;
    summary->num_cg_iterations += spent.num_cg_iterations; summary->num_residual_sweeps += spent.num_residual_sweeps; summary->num_linearizations += spent.num_linearizations;
    summary->num_graph_launches += spent.num_graph_launches; summary->num_collectives += spent.num_collectives; summary->num_pcg_collectives += spent.num_pcg_collectives;
    summary->num_pcg_launched += spent.num_pcg_launched; summary->t_linearize_ms += spent.t_linearize_ms; summary->t_sweep_ms += spent.t_sweep_ms; summary->t_cg_ms += spent.t_cg_ms;
  }
  if (st) return (gsfm_status)st;
  if (int st2 = resident ? download_state_resident(P, rot) : download_state(P, rot)) return (gsfm_status)st2;
  summary->t_total_ms = now_ms() - t0;
  return GSFM_OK;
}

// EstimateRotationsWithSigmaConsensus (estimator.cpp:314-457)
gsfm_status gsfm_rot_solve_sigma_consensus(gsfm_rot_problem* P, double* rot, int32_t iters_num, double sigma_max,
                                           const gsfm_rot_options* opt, gsfm_rot_summary* summary) {
  if (!P || !rot) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (P->error_type != GSFM_ROT_ANGLE_AXIS) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "sigma consensus needs an ANGLE_AXIS problem");
  DeviceGuard g(P->device);
  const gsfm_rot_options o = opt ? *opt : default_options();
  gsfm_rot_summary local, total; if (!summary) summary = &local;
  std::memset(&total, 0, sizeof(total));
  const double t0 = now_ms();
  const MagsacConst c = magsac_const(3);
  const double squared_sigma_max_2 = sigma_max * sigma_max * 2.0;
  const double dof_minus_one_per_two = (c.nu - 1.0) / 2.0;
  const double one_over_sigma = c.C * std::pow(2.0, dof_minus_one_per_two) / sigma_max;
  const double weight_zero = one_over_sigma * (std::tgamma(dof_minus_one_per_two) - c.gk);
  // The scalar-weight planes live on the device for the whole loop, each in its kernel's own order.  There is no weight pass: the first
  // cost sweep and the first linearisation of every inner solve start from exactly the rotations the reference computes the weights at
  // (:378-416), so they compute, store and use them (kernels.hpp, SigmaDev).  The first comparison is against zero weights, like the
  // reference's zero-initialised last_weights (:352-353).
  if (P->wmode == W_NONE) {
    if (P->cost.ws.alloc(P->cost.n) != hipSuccess || P->dir.ws.alloc(P->dir.n) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc weight planes");
    P->wmode = W_SCALAR;
  }
  if (P->cost.n) HIPCHK_S(hipMemsetAsync(P->cost.ws.p, 0, 8 * P->cost.n, P->stream));
  if (!P->sigma_table.p && P->sigma_table.upload(magsac_table(3)) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc sigma consensus buffers");
  if (!P->sigma_sum.p && P->sigma_sum.alloc(2, true) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc sigma consensus buffers");
  P->sigma.table = P->sigma_table.p; P->sigma.table_len = c.n; P->sigma.on = 0; P->sigma.ssm2 = squared_sigma_max_2; P->sigma.inv_ssm2 = 1.0 / squared_sigma_max_2;
  P->sigma.one_over_sigma = one_over_sigma; P->sigma.gk = c.gk; P->sigma.weight_zero = weight_zero;
  double global_edges = (double)P->n_edges_in;
  if (P->sharded) {
    // every edge is counted in mean |w - w_old| by exactly one rank: its cost owner
    const double mine = (double)P->cost.n;
    HIPCHK_S(hipMemcpyAsync(P->sigma_sum.p + 1, &mine, 8, hipMemcpyHostToDevice, P->stream));
    if (int st = all_reduce(P, P->sigma_sum.p + 1, 1)) return (gsfm_status)st;
    HIPCHK_S(hipMemcpyAsync(&global_edges, P->sigma_sum.p + 1, 8, hipMemcpyDeviceToHost, P->stream));
    if (int st = sync_check(P, "sigma consensus: edge count")) return (gsfm_status)st;
  }
  int outer = 0;
  for (int it = 0; it < iters_num; ++it) {
    ++outer;
    P->sigma_pending_cost = P->sigma_pending_lin = true;   // consumed by the solve's first K1 / K2
    const gsfm_status sst = gsfm_rot_solve(P, rot, &o, summary);
    P->sigma_pending_cost = P->sigma_pending_lin = false;
    if (sst) return sst;
    // sum |w - w_old| was left on the device by that first sweep: read it now, after the solve (the decision it feeds comes after the
    // solve in the reference too, :448)
    if (int st = all_reduce(P, P->sigma_sum.p, 1)) return (gsfm_status)st;
    double avg = 0.0;
    if (hipMemcpyAsync(&avg, P->sigma_sum.p, 8, hipMemcpyDeviceToHost, P->stream) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "read weight change");
    if (int st = sync_check(P, "sigma consensus weights")) return (gsfm_status)st;
    avg /= global_edges;
    if (it == 0) total = *summary;
    else {
      total.num_iterations += summary->num_iterations; total.num_successful_steps += summary->num_successful_steps;
      total.num_unsuccessful_steps += summary->num_unsuccessful_steps; total.num_residual_sweeps += summary->num_residual_sweeps;
      total.num_linearizations += summary->num_linearizations; total.num_cg_iterations += summary->num_cg_iterations;
      total.final_cost = summary->final_cost; total.termination = summary->termination;
      total.final_gradient_max_norm = summary->final_gradient_max_norm; total.final_radius = summary->final_radius;
      total.num_dense_solves += summary->num_dense_solves; total.num_graph_launches += summary->num_graph_launches;
      total.num_collectives += summary->num_collectives; total.num_pcg_collectives += summary->num_pcg_collectives; total.num_pcg_launched += summary->num_pcg_launched;
      total.t_linearize_ms += summary->t_linearize_ms; total.t_sweep_ms += summary->t_sweep_ms; total.t_cg_ms += summary->t_cg_ms;
      total.num_forcing_refinements += summary->num_forcing_refinements; total.num_inexact_steps += summary->num_inexact_steps;
      total.num_pcg_capped_steps += summary->num_pcg_capped_steps; total.num_forcing_restarts += summary->num_forcing_restarts;
      total.worst_accepted_cg_residual = std::fmax(total.worst_accepted_cg_residual, summary->worst_accepted_cg_residual);
    }
    total.last_weight_change = avg;
    if (avg <= 1e-7) break;  // :448
  }
  total.outer_iterations = outer; total.t_total_ms = now_ms() - t0;
  *summary = total;
  return GSFM_OK;
}

gsfm_status gsfm_rot_residuals(gsfm_rot_problem* P, const double* rot, double* s_out, double* rho_out, double* r_out, double* cost) {
  if (!P || !rot) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  const size_t E = P->n_edges_in, Ec = P->cost.n, R = (size_t)P->res_dim;
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  // The sweep writes in the problem's own edge order (coalesced); the caller's order is restored here, at the C-ABI edge, on the host:
  // out[edge_order[u]] = device[u].  Edges this rank does not count in the cost (sharded problems) stay zero.
  DevBuf<double> dr; DevBuf<double2> dsr, dr12;
  if (((s_out || rho_out) && dsr.alloc(Ec) != hipSuccess) || (rho_out && dr12.alloc(Ec) != hipSuccess) || (r_out && dr.alloc(R * Ec) != hipSuccess))
    return (gsfm_status)fail(GSFM_ERR_HIP, "alloc outputs");
  CostOutputs out;
  out.srho = dsr.p; out.rho12 = dr12.p; out.r = dr.p;
  if (int st = launch_cost(P, P->q.p, SC_COST, out)) return (gsfm_status)st;
  double h[SC_N];
  if (int st = read_scalars(P, h)) return (gsfm_status)st;
  if (cost) *cost = h[SC_COST];
  const std::vector<uint32_t>& ord = P->h_cost_eid;
  std::vector<double> stage;
  if (s_out || rho_out) {   // device planes: (s, rho) and (rho', rho'') per edge
    stage.resize(4 * Ec);
    if (Ec && (hipMemcpy(stage.data(), dsr.p, 16 * Ec, hipMemcpyDeviceToHost) != hipSuccess || (rho_out && hipMemcpy(stage.data() + 2 * Ec, dr12.p, 16 * Ec, hipMemcpyDeviceToHost) != hipSuccess)))
      return (gsfm_status)fail(GSFM_ERR_HIP, "copy s / rho");
    if (s_out) { std::memset(s_out, 0, 8 * E); for (size_t u = 0; u < Ec; ++u) s_out[ord[u]] = stage[2 * u]; }
    if (rho_out) {
      std::memset(rho_out, 0, 24 * E);
      for (size_t u = 0; u < Ec; ++u) { double* o = rho_out + 3 * (size_t)ord[u]; o[0] = stage[2 * u + 1]; o[1] = stage[2 * Ec + 2 * u]; o[2] = stage[2 * Ec + 2 * u + 1]; }
    }
  }
  if (r_out) {
    stage.resize(R * Ec);
    if (Ec && hipMemcpy(stage.data(), dr.p, 8 * R * Ec, hipMemcpyDeviceToHost) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "copy r");
    std::memset(r_out, 0, 8 * R * E);
    for (size_t u = 0; u < Ec; ++u) for (size_t k = 0; k < R; ++k) r_out[R * (size_t)ord[u] + k] = stage[k * Ec + u];
  }
  return GSFM_OK;
}

int64_t gsfm_rot_edge_order(gsfm_rot_problem* P, uint32_t* order_out, uint64_t cap) {
  if (!P) { fail(GSFM_ERR_INVALID_ARG, "NULL problem"); return -1; }
  const size_t n = P->h_cost_eid.size();
  if (order_out) std::memcpy(order_out, P->h_cost_eid.data(), 4 * std::min<size_t>(n, cap));
  return (int64_t)n;
}

gsfm_status gsfm_rot_linearize(gsfm_rot_problem* P, const double* rot, double* gradient, double* diag_blocks, double* cost) {
  if (!P || !rot) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  const size_t N = P->n_cams;
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  if (int st = launch_cost(P, P->q.p, SC_COST)) return (gsfm_status)st;
  if (int st = launch_lin(P, P->q.p)) return (gsfm_status)st;
  DevBuf<double> dg, dblk;
  if (dg.alloc(3 * N) != hipSuccess || dblk.alloc(9 * N) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc outputs");
  hipLaunchKernelGGL(k_cam_export, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->gD.p, P->n_cams, P->param_dim, dg.p, dblk.p, P->D6.p);
  double h[SC_N];
  if (int st = read_scalars(P, h)) return (gsfm_status)st;
  if (cost) *cost = h[SC_COST];
  std::vector<double> stage;
  if (!P->perm.empty()) stage.resize(9 * N);
  if (gradient) {
    if (hipMemcpy(P->perm.empty() ? gradient : stage.data(), dg.p, 24 * N, hipMemcpyDeviceToHost) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "copy gradient");
    if (!P->perm.empty()) to_external(P, stage.data(), gradient, 3);
  }
  if (diag_blocks) {
    if (hipMemcpy(P->perm.empty() ? diag_blocks : stage.data(), dblk.p, 72 * N, hipMemcpyDeviceToHost) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "copy blocks");
    if (!P->perm.empty()) to_external(P, stage.data(), diag_blocks, 9);
  }
  return GSFM_OK;
}

gsfm_status gsfm_rot_normal_matvec(gsfm_rot_problem* P, const double* v, double* y) {
  if (!P || !v || !y) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (!P->have_lin) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "call gsfm_rot_linearize first");
  DeviceGuard g(P->device);
  const size_t N = P->n_cams;
  // y = T^T B_eta (T v): xcg <- v, p <- T v, Ap <- B p, xcg <- T^T Ap
  if (hipMemcpyAsync(P->xcg.p, to_internal(P, v, 3), 24 * N, hipMemcpyHostToDevice, P->stream) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "upload v");
  hipLaunchKernelGGL(k_cam_apply_T, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->n_cams, P->param_dim, 0, P->xcg.p, P->p.p);
  if (P->lin_is_lap) hipLaunchKernelGGL(k_cam_rotT, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, (const double*)P->p.p, P->q_lin, P->n_cams, P->u_rot.p);
  if (int st = launch_matvec(P, P->D6.p, P->p.p, P->Ap.p, nullptr)) return (gsfm_status)st;
  hipLaunchKernelGGL(k_cam_apply_T, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->n_cams, P->param_dim, 1, P->Ap.p, P->xcg.p);
  double* dst = y;
  if (!P->perm.empty()) { P->h_cam.resize(3 * N); dst = P->h_cam.data(); }
  if (hipMemcpyAsync(dst, P->xcg.p, 24 * N, hipMemcpyDeviceToHost, P->stream) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "download y");
  if (int st = sync_check(P, "normal_matvec")) return (gsfm_status)st;
  if (!P->perm.empty()) to_external(P, dst, y, 3);
  return GSFM_OK;
}

gsfm_status gsfm_rot_loss_eval(gsfm_rot_problem* P, const double* s, uint64_t n, double* rho3_out, double* value_out, double* rho1_fast_out) {
  if (!P || (n > 0 && !s)) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (P->cb) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "loss_eval needs a native loss program (a host-callback loss runs on the host)");
  if (n == 0) return GSFM_OK;
  DeviceGuard g(P->device);
  DevBuf<double> ds, d3, dv, d1;
  const int lm = loss_mode(P);
  if (rho1_fast_out && lm == LM_PROGRAM) { for (uint64_t k = 0; k < n; ++k) rho1_fast_out[k] = std::numeric_limits<double>::quiet_NaN(); rho1_fast_out = nullptr; }   // K2 has no fast path for this program
  if (ds.alloc(n) != hipSuccess || (rho3_out && d3.alloc(3 * n) != hipSuccess) || (value_out && dv.alloc(n) != hipSuccess) || (rho1_fast_out && d1.alloc(n) != hipSuccess)) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc");
  HIPCHK_S(hipMemcpyAsync(ds.p, s, 8 * n, hipMemcpyHostToDevice, P->stream));
  const dim3 grid(grid_for(n)), blk(GSFM_BLOCK);
  switch (lm) {   // the specialisation K1 / K2 are dispatched on for this program
    case LM_SIMPLE: hipLaunchKernelGGL(k_loss_eval<LM_SIMPLE>, grid, blk, 0, P->stream, (const DevLoss*)P->d_loss.p, (const double*)ds.p, (size_t)n, d3.p, dv.p, d1.p); break;
    case LM_MAGSAC: hipLaunchKernelGGL(k_loss_eval<LM_MAGSAC>, grid, blk, 0, P->stream, (const DevLoss*)P->d_loss.p, (const double*)ds.p, (size_t)n, d3.p, dv.p, d1.p); break;
    default: hipLaunchKernelGGL(k_loss_eval<LM_PROGRAM>, grid, blk, 0, P->stream, (const DevLoss*)P->d_loss.p, (const double*)ds.p, (size_t)n, d3.p, dv.p, d1.p); break;
  }
  if (rho1_fast_out) HIPCHK_S(hipMemcpyAsync(rho1_fast_out, d1.p, 8 * n, hipMemcpyDeviceToHost, P->stream));
  if (rho3_out) HIPCHK_S(hipMemcpyAsync(rho3_out, d3.p, 24 * n, hipMemcpyDeviceToHost, P->stream));
  if (value_out) HIPCHK_S(hipMemcpyAsync(value_out, dv.p, 8 * n, hipMemcpyDeviceToHost, P->stream));
  return (gsfm_status)sync_check(P, "loss_eval");
}

gsfm_status gsfm_rot_edge_sq_norms(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j, const double* rel_aa,
                                   const double* cov6, const double* rot_aa, double max_sq_norm, double* s_out, uint8_t* keep_out,
                                   uint64_t* n_kept, double* kernel_ms) {
  if (n_cams == 0 || n_edges == 0) { if (n_kept) *n_kept = 0; return GSFM_OK; }
  if (!edge_i || !edge_j || !rel_aa || !rot_aa || !s_out) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  for (uint64_t e = 0; e < n_edges; ++e) if (edge_i[e] >= n_cams || edge_j[e] >= n_cams) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "edge with an out-of-range camera index");
  // (no host fallback: like every entry point of this library the sweep runs on the device or fails loudly)
  if (const char* why = no_device_reason("the edge sweep")) return (gsfm_status)fail(GSFM_ERR_NO_DEVICE, why);
  // One device slab, one private stream, stream-ordered copies: no hipDeviceSynchronize (it would stall every problem's stream of the
  // process) and nothing to leak on an error path (the guard below owns stream, events and slab).
  struct Guard {
    hipStream_t s = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; void* slab = nullptr;
    ~Guard() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); if (slab) (void)hipFree(slab); if (s) (void)hipStreamDestroy(s); }
  } G;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t o_i = 0, o_j = o_i + up(4 * n_edges), o_rel = o_j + up(4 * n_edges), o_cov = o_rel + up(24 * n_edges), o_rot = o_cov + (cov6 ? up(48 * n_edges) : 0),
               o_s = o_rot + up(24 * (size_t)n_cams), o_q = o_s + up(8 * n_edges), o_keep = o_q + up(32 * (size_t)n_cams), o_cnt = o_keep + (keep_out ? up(n_edges) : 0), total = o_cnt + 256;
  HIPCHK_S(hipStreamCreateWithFlags(&G.s, hipStreamNonBlocking));
  HIPCHK_S(hipEventCreate(&G.e0)); HIPCHK_S(hipEventCreate(&G.e1));
  if (hipMalloc(&G.slab, total) != hipSuccess) { G.slab = nullptr; return (gsfm_status)fail(GSFM_ERR_HIP, "allocating the edge sweep buffers failed"); }
  char* base = (char*)G.slab;
  HIPCHK_S(hipMemcpyAsync(base + o_i, edge_i, 4 * n_edges, hipMemcpyHostToDevice, G.s)); HIPCHK_S(hipMemcpyAsync(base + o_j, edge_j, 4 * n_edges, hipMemcpyHostToDevice, G.s));
  HIPCHK_S(hipMemcpyAsync(base + o_rel, rel_aa, 24 * n_edges, hipMemcpyHostToDevice, G.s)); HIPCHK_S(hipMemcpyAsync(base + o_rot, rot_aa, 24 * (size_t)n_cams, hipMemcpyHostToDevice, G.s));
  if (cov6) HIPCHK_S(hipMemcpyAsync(base + o_cov, cov6, 48 * n_edges, hipMemcpyHostToDevice, G.s));
  HIPCHK_S(hipMemsetAsync(base + o_cnt, 0, 8, G.s));
  hipLaunchKernelGGL(k_cam_cache, dim3(grid_for(n_cams)), dim3(GSFM_BLOCK), 0, G.s, (const double*)(base + o_rot), n_cams, 3, (double2*)(base + o_q));
  EdgeSweepArgs a{};
  a.n = n_edges; a.ei = (const uint32_t*)(base + o_i); a.ej = (const uint32_t*)(base + o_j); a.rel_aa = (const double*)(base + o_rel); a.cov6 = cov6 ? (const double*)(base + o_cov) : nullptr;
  a.q = (const double2*)(base + o_q); a.max_sq = max_sq_norm; a.s_out = (double*)(base + o_s); a.keep = keep_out ? (uint8_t*)(base + o_keep) : nullptr; a.n_kept = (unsigned long long*)(base + o_cnt);
  HIPCHK_S(hipEventRecord(G.e0, G.s));
  hipLaunchKernelGGL(k_edge_sweep, dim3(grid_for(n_edges)), dim3(GSFM_BLOCK), 0, G.s, a);
  HIPCHK_S(hipEventRecord(G.e1, G.s));
  HIPCHK_S(hipMemcpyAsync(s_out, base + o_s, 8 * n_edges, hipMemcpyDeviceToHost, G.s));
  if (keep_out) HIPCHK_S(hipMemcpyAsync(keep_out, base + o_keep, n_edges, hipMemcpyDeviceToHost, G.s));
  unsigned long long cnt = 0;
  HIPCHK_S(hipMemcpyAsync(&cnt, base + o_cnt, 8, hipMemcpyDeviceToHost, G.s));
  HIPCHK_S(hipStreamSynchronize(G.s));
  HIPCHK_S(hipGetLastError());
  float ms = 0; (void)hipEventElapsedTime(&ms, G.e0, G.e1);
  if (kernel_ms) *kernel_ms = ms;
  if (n_kept) *n_kept = keep_out ? (uint64_t)cnt : n_edges;
  return GSFM_OK;
}

int64_t gsfm_rot_count_components(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j) {
  if ((n_edges > 0 && (!edge_i || !edge_j)) || n_cams >= 0x7fffffffu) { fail(GSFM_ERR_INVALID_ARG, "bad argument"); return -1; }
  for (uint64_t e = 0; e < n_edges; ++e) if (edge_i[e] >= n_cams || edge_j[e] >= n_cams) { fail(GSFM_ERR_INVALID_ARG, "edge with an out-of-range camera index"); return -1; }
  return (int64_t)count_components(n_cams, n_edges, edge_i, edge_j);
}

int32_t gsfm_rot_locality_order(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j, uint32_t* perm_out) {
  if (!perm_out || (n_edges > 0 && (!edge_i || !edge_j)) || n_cams >= 0x7fffffffu || n_edges >= 0x7fffffffull) { fail(GSFM_ERR_INVALID_ARG, "bad argument"); return -1; }
  std::vector<uint32_t> ptr((size_t)n_cams + 1, 0), adj(2 * n_edges);
  for (uint64_t e = 0; e < n_edges; ++e) {
    if (edge_i[e] >= n_cams || edge_j[e] >= n_cams || edge_i[e] == edge_j[e]) { fail(GSFM_ERR_INVALID_ARG, "edge with an out-of-range or repeated camera index"); return -1; }
    ptr[edge_i[e] + 1]++; ptr[edge_j[e] + 1]++;
  }
  for (size_t c = 0; c < n_cams; ++c) ptr[c + 1] += ptr[c];
  {
    std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
    for (uint64_t e = 0; e < n_edges; ++e) { adj[fill[edge_i[e]]++] = edge_j[e]; adj[fill[edge_j[e]]++] = edge_i[e] | 0x80000000u; }
  }
  std::vector<uint32_t> perm;
  const bool adopted = n_edges > 0 && reorder_for_locality(n_cams, n_edges, edge_i, edge_j, ptr, adj, &perm);
  for (uint32_t c = 0; c < n_cams; ++c) perm_out[c] = adopted ? perm[c] : c;
  return adopted ? 1 : 0;
}

int32_t gsfm_rot_get_trace(gsfm_rot_problem* P, double* out, int32_t cap_rows) {
  if (!P) return 0;
  const int rows = (int)(P->trace.size() / GSFM_ROT_TRACE_COLS);
  const int m = std::min(rows, cap_rows);
  if (out && m > 0) std::memcpy(out, P->trace.data(), sizeof(double) * GSFM_ROT_TRACE_COLS * m);
  return rows;
}

gsfm_status gsfm_rot_time_sweep(gsfm_rot_problem* P, const double* rot, int32_t reps, double* mean_ms) {
  if (!P || !rot || !mean_ms || reps <= 0) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "bad argument");
  if (P->cb) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "time_sweep needs a native loss");
  DeviceGuard g(P->device);
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  const CostArgs a = cost_args(P, P->q.p);
  for (int k = 0; k < 3; ++k) if (dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost)) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "dispatch");
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "event create");
  (void)hipEventRecord(e0, P->stream);
  for (int k = 0; k < reps; ++k) dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost);
  (void)hipEventRecord(e1, P->stream);
  int st = sync_check(P, "time_sweep");
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *mean_ms = ms / reps;
  return (gsfm_status)st;
}

gsfm_status gsfm_rot_time_sweep_variants(gsfm_rot_problem* P, const double* rot, int32_t reps, double* out_ms8) {
  if (!P || !rot || !out_ms8 || reps <= 0) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "bad argument");
  if (P->cb) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "time_sweep_variants needs a native loss");
  DeviceGuard g(P->device);
  const size_t Ec = P->cost.n;
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  DevBuf<double> ds; DevBuf<double2> dsr, dr12;
  if (ds.alloc(Ec, true) != hipSuccess || dsr.alloc(Ec, true) != hipSuccess || dr12.alloc(Ec, true) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc outputs");
  const CostArgs base = cost_args(P, P->q.p);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "event create");
  auto timed = [&](auto&& launch, double* out) -> int {   // two rounds, the faster one counts (a round now and then is hit by something else on the box)
    double best = 0.0;
    for (int round = 0; round < 2; ++round) {
      for (int k = -2; k < reps; ++k) { if (k == 0) (void)hipEventRecord(e0, P->stream); launch(); }
      (void)hipEventRecord(e1, P->stream);
      if (int st = sync_check(P, "time_sweep_variants")) return st;
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      if (round == 0 || ms / reps < best) best = ms / reps;
    }
    *out = best; return 0;
  };
  int st = 0;
  for (int k = 0; k < 8; ++k) out_ms8[k] = 0.0;
  { CostArgs a = base; st = timed([&] { dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost); }, &out_ms8[0]); }                                   // trial cost: rho value only
  if (!st) { CostArgs a = base; a.srho_out = dsr.p; a.rho12_out = dr12.p; st = timed([&] { dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost); }, &out_ms8[1]); }  // s + rho triple
  if (!st) { CostArgs a = base; a.s_out = ds.p; a.s_only = 1; st = timed([&] { dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost); }, &out_ms8[2]); }   // s only (callback pass 1)
  if (!st) { CostArgs a = base; a.rho1_out = ds.p; st = timed([&] { dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost); }, &out_ms8[3]); }             // the reweight sweep of SURVEY 8(d): rho' out
  if (!st && P->functor == F_AA && P->wmode == W_SCALAR && P->cost.ws.p && P->dir.ws.p) {
    // sigma consensus: the first cost sweep / linearisation of an inner solve with the weight computation fused in (estimator.cpp:400-416),
    // against the same kernels without it.  The sweeps overwrite the problem's own weight planes: save and restore them.
    const MagsacConst c = magsac_const(3);
    if (!P->sigma_table.p && P->sigma_table.upload(magsac_table(3)) != hipSuccess) st = fail(GSFM_ERR_HIP, "alloc");
    if (!st && !P->sigma_sum.p && P->sigma_sum.alloc(2, true) != hipSuccess) st = fail(GSFM_ERR_HIP, "alloc");
    DevBuf<double> keep_c, keep_d;
    if (!st && (keep_c.alloc(P->cost.n) != hipSuccess || keep_d.alloc(P->dir.n) != hipSuccess)) st = fail(GSFM_ERR_HIP, "alloc");
    if (!st) {
      const double sigma_max = 0.02, one_over_sigma = c.C * std::pow(2.0, (c.nu - 1.0) / 2.0) / sigma_max;
      SigmaDev sg{};
      sg.table = P->sigma_table.p; sg.table_len = c.n; sg.on = 1; sg.ssm2 = 2.0 * sigma_max * sigma_max; sg.inv_ssm2 = 1.0 / sg.ssm2; sg.one_over_sigma = one_over_sigma;
      sg.gk = c.gk; sg.weight_zero = one_over_sigma * (std::tgamma((c.nu - 1.0) / 2.0) - c.gk);
      (void)hipMemcpyAsync(keep_c.p, P->cost.ws.p, 8 * P->cost.n, hipMemcpyDeviceToDevice, P->stream);
      (void)hipMemcpyAsync(keep_d.p, P->dir.ws.p, 8 * P->dir.n, hipMemcpyDeviceToDevice, P->stream);
      const SigmaDev keep_sigma = P->sigma;
      P->sigma = sg;
      st = timed([&] { P->sigma_pending_cost = true; (void)launch_cost(P, P->q.p, SC_COST); }, &out_ms8[4]);    // K1 with the weights fused in (+ the two scalar reductions)
      if (!st) st = timed([&] { (void)launch_cost(P, P->q.p, SC_COST); }, &out_ms8[5]);                          // the same sweep without
      if (!st) st = timed([&] { P->sigma_pending_lin = true; (void)launch_lin(P, P->q.p); }, &out_ms8[6]);       // K2 with the weights fused in
      if (!st) st = timed([&] { (void)launch_lin(P, P->q.p); }, &out_ms8[7]);                                    // K2 without
      P->sigma = keep_sigma; P->sigma_pending_cost = P->sigma_pending_lin = false;
      (void)hipMemcpyAsync(P->cost.ws.p, keep_c.p, 8 * P->cost.n, hipMemcpyDeviceToDevice, P->stream);
      (void)hipMemcpyAsync(P->dir.ws.p, keep_d.p, 8 * P->dir.n, hipMemcpyDeviceToDevice, P->stream);
      if (int s2 = sync_check(P, "time_sweep_variants restore")) st = st ? st : s2;
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (gsfm_status)st;
}

gsfm_status gsfm_rot_time_kernels(gsfm_rot_problem* P, const double* rot, int32_t reps, double* out_ms4) {
  if (!P || !rot || !out_ms4 || reps <= 0) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "bad argument");
  if (P->cb) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "time_kernels needs a native loss");
  DeviceGuard g(P->device);
  const gsfm_rot_options o = default_options();
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  if (int st = launch_lin(P, P->q.p)) return (gsfm_status)st;
  launch_prep(P, o, o.initial_trust_region_radius, true);
  if (int st = sync_check(P, "time_kernels setup")) return (gsfm_status)st;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "event create");
  for (int which = 0; which < 4; ++which) {
    out_ms4[which] = 0.0;
    if (which == 3) continue;   // (reserved)
    for (int round = 0; round < 2; ++round) {   // two rounds of `reps` launches each, the faster round's mean counts
      for (int k = -2; k < reps; ++k) {  // two warm-up launches
        if (k == 0) (void)hipEventRecord(e0, P->stream);
        if (which == 0) { if (int st = launch_cost(P, P->q.p, SC_COST)) return (gsfm_status)st; }
        else if (which == 1) { if (int st = launch_lin(P, P->q.p)) return (gsfm_status)st; }
        else if (which == 2) { if (int st = launch_matvec(P, P->Mblk.p, P->b.p, P->Ap.p, nullptr)) return (gsfm_status)st; }
      }
      (void)hipEventRecord(e1, P->stream);
      if (int st = sync_check(P, "time_kernels")) return (gsfm_status)st;
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (round == 0 || ms / reps < out_ms4[which]) out_ms4[which] = ms / reps;
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return GSFM_OK;
}

gsfm_status gsfm_rot_matvec_bytes(gsfm_rot_problem* P, double* layout_bytes, double* lin_bytes, int32_t* form) {
  if (!P) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL problem");
  const double N = (double)P->n_rows;
  const bool lap = P->lap_capable;
  if (lap && P->cs.active) {
    // mat-vec, per position: its 2-, 4- or 6-byte record (column | slot | row count; 2: delta-coded, ColLayoutDev::k16) + body-frame block 48; partial sums written and
    // read once; the gathered vector, the diagonal blocks, p, q in, y out once per camera.  Linearisation, per position: record 8 +
    // q_rel 32 + whitening 48 in, block 48 out; nine partial sums per row and workgroup
    if (layout_bytes) *layout_bytes = (P->cs.k16_active ? 50.0 : P->cs.cmax ? 52.0 : 54.0) * (double)P->cs.n_pos + 2.0 * 24.0 * P->cs.n_wg * GSFM_COL_RB + (24.0 + 48.0 + 24.0 + 32.0 + 24.0) * N;
    if (lin_bytes) *lin_bytes = (8.0 + (P->q3 ? 24.0 : 32.0) + (P->wmode == W_MATRIX ? 48.0 : P->wmode == W_SCALAR ? 8.0 : 0.0) + 48.0) * (double)P->cs.n_pos + 2.0 * 72.0 * P->cs.n_wg * GSFM_COL_RB + (32.0 + 72.0) * N;
    if (form) *form = 2;
  } else {
    if (layout_bytes) *layout_bytes = (double)P->dir.n * (lap ? 52.0 : 76.0) + 2.0 * 24.0 * N;
    const double w = P->wmode == W_MATRIX ? 48.0 : P->wmode == W_SCALAR ? 8.0 : 0.0;
    if (lin_bytes) *lin_bytes = (double)P->dir.n * (4.0 + (P->q3 ? 24.0 : 32.0) + w + (lap ? 48.0 : 72.0)) + (32.0 + 72.0) * N;
    if (form) *form = lap ? 1 : 0;
  }
  return GSFM_OK;
}

gsfm_status gsfm_rot_sweep_bytes(gsfm_rot_problem* P, double* algorithmic, double* layout) {
  if (!P) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL problem");
  // indices 8 B + measurement 24 B (32 B for the quaternion types) + whitening 48/8/0 B + weight out 8 B
  const double w = P->wmode == W_MATRIX ? 48.0 : P->wmode == W_SCALAR ? 8.0 : 0.0;
  if (algorithmic) *algorithmic = 8.0 + (P->functor == F_AA ? 24.0 : 32.0) + w + 8.0;
  // as laid out: uint2 idx + the measurement (24 B on the W_MATRIX problems of >= 1 M edges since round 6: three quaternion components, kernels.hpp qrel_three; 32 B otherwise) + whitening planes; the weight is consumed in-kernel (no per-edge store)
  if (layout) *layout = 8.0 + (P->q3 ? 24.0 : 32.0) + w;
  return GSFM_OK;
}

gsfm_status gsfm_cov_estimate(uint64_t n_edges, const uint64_t* match_ptr, const double* matches, const double* intrinsics,
                              const double* rot_in, const double* trans_in, int32_t max_iterations, double* cov9_out,
                              double* rot_out, double* trans_out, int32_t* status_out, int32_t* iters_out, double* kernel_ms) {
  if (!match_ptr || !matches || !intrinsics || !rot_in || !trans_in || !cov9_out || !rot_out || !trans_out || !status_out)
    return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (n_edges == 0) return GSFM_OK;
  if (const char* why = no_device_reason("the covariance estimator")) return (gsfm_status)fail(GSFM_ERR_NO_DEVICE, why);
  const uint64_t n_matches = match_ptr[n_edges];
  for (uint64_t e = 0; e < n_edges; ++e) if (match_ptr[e + 1] < match_ptr[e]) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "match_ptr must be non-decreasing");
  DevBuf<uint64_t> d_ptr; DevBuf<double4> d_m; DevBuf<double> d_K, d_r, d_t, d_cov, d_ro, d_to; DevBuf<int> d_st, d_it;
  bool ok = d_ptr.alloc(n_edges + 1) == hipSuccess && d_m.alloc(std::max<uint64_t>(n_matches, 1)) == hipSuccess && d_K.alloc(6 * n_edges) == hipSuccess &&
            d_r.alloc(3 * n_edges) == hipSuccess && d_t.alloc(3 * n_edges) == hipSuccess && d_cov.alloc(9 * n_edges) == hipSuccess &&
            d_ro.alloc(3 * n_edges) == hipSuccess && d_to.alloc(3 * n_edges) == hipSuccess && d_st.alloc(n_edges) == hipSuccess && d_it.alloc(n_edges) == hipSuccess;
  if (!ok) return (gsfm_status)fail(GSFM_ERR_HIP, "allocating covariance buffers failed");
  HIPCHK_S(hipMemcpy(d_ptr.p, match_ptr, 8 * (n_edges + 1), hipMemcpyHostToDevice));
  if (n_matches) HIPCHK_S(hipMemcpy(d_m.p, matches, 32 * n_matches, hipMemcpyHostToDevice));
  HIPCHK_S(hipMemcpy(d_K.p, intrinsics, 48 * n_edges, hipMemcpyHostToDevice));
  HIPCHK_S(hipMemcpy(d_r.p, rot_in, 24 * n_edges, hipMemcpyHostToDevice));
  HIPCHK_S(hipMemcpy(d_t.p, trans_in, 24 * n_edges, hipMemcpyHostToDevice));
  CovArgs a{};
  a.n_edges = n_edges; a.match_ptr = d_ptr.p; a.matches = d_m.p; a.intr = d_K.p; a.rot_in = d_r.p; a.trans_in = d_t.p;
  a.max_iterations = max_iterations; a.cov9 = d_cov.p; a.rot_out = d_ro.p; a.trans_out = d_to.p; a.status = d_st.p; a.iters = d_it.p;
  hipEvent_t e0, e1;
  HIPCHK_S(hipEventCreate(&e0)); HIPCHK_S(hipEventCreate(&e1));
  const int grid = (int)((n_edges + (GSFM_BLOCK / 64) - 1) / (GSFM_BLOCK / 64));
  HIPCHK_S(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_cov_estimate, dim3(grid), dim3(GSFM_BLOCK), 0, 0, a);
  HIPCHK_S(hipEventRecord(e1, 0));
  HIPCHK_S(hipDeviceSynchronize());
  HIPCHK_S(hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (kernel_ms) *kernel_ms = ms;
  HIPCHK_S(hipMemcpy(cov9_out, d_cov.p, 72 * n_edges, hipMemcpyDeviceToHost));
  HIPCHK_S(hipMemcpy(rot_out, d_ro.p, 24 * n_edges, hipMemcpyDeviceToHost));
  HIPCHK_S(hipMemcpy(trans_out, d_to.p, 24 * n_edges, hipMemcpyDeviceToHost));
  HIPCHK_S(hipMemcpy(status_out, d_st.p, 4 * n_edges, hipMemcpyDeviceToHost));
  if (iters_out) HIPCHK_S(hipMemcpy(iters_out, d_it.p, 4 * n_edges, hipMemcpyDeviceToHost));
  return GSFM_OK;
}

int32_t gsfm_magsac_table(int32_t nu, double* out, int32_t cap) {
  if (nu != 3 && nu != 4 && nu != 9) return -1;
  const std::vector<double>& t = magsac_table(nu);
  const int m = std::min((int)t.size(), cap);
  if (out && m > 0) std::memcpy(out, t.data(), 8 * (size_t)m);
  return (int)t.size();
}
gsfm_status gsfm_magsac_constants(int32_t nu, double* C, double* q, double* gk) {
  if (nu != 3 && nu != 4 && nu != 9) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "nu must be 3, 4 or 9");
  const MagsacConst c = magsac_const(nu);
  if (C) *C = c.C; if (q) *q = c.q; if (gk) *gk = c.gk;
  return GSFM_OK;
}

}  // extern "C"
