// Host side, part 5: Levenberg-Marquardt control with the semantics of Ceres 1.14's TrustRegionMinimizer / LevenbergMarquardtStrategy
// (the reference's ceres::Solve calls, src/GSfM_nonlinear_rotation_estimator.cpp:77,182,305,446) and the state transfers around it.
#pragma once
#include "host_common.hpp"

namespace gsfm {
// Per-camera 3-vectors between the caller's numbering and the internal one (the locality relabelling adopted at create), on the device:
// to_internal: dst[perm[k]] = src[k]; otherwise dst[k] = src[perm[k]].  (gsfm_rot_solve_resident; the host entry points permute on the host.)
__global__ void __launch_bounds__(GSFM_BLOCK) k_cam_permute3(const double* __restrict__ src, const uint32_t* __restrict__ perm, uint32_t n, int to_internal, double* __restrict__ dst) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k >= n) return;
  const size_t a = to_internal ? k : perm[k], b = to_internal ? perm[k] : k;
  dst[3 * b] = src[3 * a]; dst[3 * b + 1] = src[3 * a + 1]; dst[3 * b + 2] = src[3 * a + 2];
}
}  // namespace gsfm

namespace {

// lm_solve's private verdict "the forcing schedule must not be trusted on this trajectory: redo the solve from the initial rotations with every
// step exact" (gsfm_rot_solve does; never leaves the library)
constexpr int GSFM_INTERNAL_RESTART = 1000;

// reduce = false: the caller sums k_cam_step's partials itself (k_lm_decide)
int launch_step(gsfm_rot_problem* P, bool inexact = false, bool reduce = true) {
  StepArgs a{};
  if (inexact && P->lap_capable) {   // (functors whose cost depends on R_j R_i^T alone: for those the gauge is an exact symmetry) a loose PCG iterate: its gauge component is taken out first (kernels.hpp, k_gauge_part) -- inside k_cam_step, the PCG state stays resumable
    const double2* gq = P->q_lin ? P->q_lin : P->q.p;
    GaugeArgs ga{P->n_cams, P->nb_cam, P->active.p, gq, P->Lam.p, P->xcg.p, P->r.p, P->part_gauge.p};
    hipLaunchKernelGGL(k_gauge_part, dim3(P->nb_cam), dim3(GSFM_BLOCK), 0, P->stream, ga);
    a.gauge_part = P->part_gauge.p; a.gauge_nb = P->nb_cam; a.gauge_q = gq;
  }
  a.Minv = inexact ? P->Minv.p : nullptr;
  a.n = P->n_cams; a.param_dim = P->param_dim; a.x = P->x.p; a.active = P->active.p; a.eta = P->xcg.p; a.b = P->b.p; a.rcg = P->r.p;
  a.Lam = P->Lam.p; a.Tinv = P->Tinv.p; a.x_trial = P->x_trial.p; a.q_trial = P->q_trial.p; a.partials = P->part_cam.p;
  hipLaunchKernelGGL(k_cam_step, dim3(P->nb_cam), dim3(GSFM_BLOCK), 0, P->stream, a);
  if (reduce) {
    hipLaunchKernelGGL(k_sum_partials_multi, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cam.p, P->nb_cam, 5, P->scal.p + SC_STEP);
    if (inexact) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cam.p + (size_t)5 * P->nb_cam, P->nb_cam, P->scal.p + SC_ZL8);
  }
  return 0;
}

// The scalar block of the LM loop (trial cost, step sums, gradient norm): through the mailbox the PCG's status takes (solver_pcg.hpp) where
// there is one -- a one-lane kernel behind the sweep's reduction and a polled stamp instead of a blit, a stream synchronisation and the way
// back -- otherwise a read-back.
int read_scalars(gsfm_rot_problem* P, double* h) {
  static_assert(SC_N <= GSFM_MAIL_WORDS, "the scalar block fits the mailbox");
  if (mail_usable(P)) {
    mail_post(P, P->scal.p, SC_N * sizeof(double));
    P->mail_expected += 1.0;
    return mail_wait(P, h, SC_N * sizeof(double));
  }
  return read_back(P, h, P->scal.p, SC_N * sizeof(double), "read scalars");
}


// The trial point of a host-controlled step: step, cost sweep and the look at the scalars.  With the mailbox (unsharded, native loss) the
// three single-workgroup kernels behind the sweep -- two reductions and the post -- are one (k_trial_post).
int evaluate_trial(gsfm_rot_problem* P, bool inexact, double* h) {
  if (!P->sharded && !P->cb && !P->sigma_pending_cost && mail_usable(P)) {
    launch_step(P, inexact, false);
    if (int st = launch_cost(P, P->q_trial.p, SC_TRIAL, CostOutputs(), false)) return st;
    hipLaunchKernelGGL(k_trial_post, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->scal.p, (int)SC_STEP, (int)SC_TRIAL, (const double*)P->part_cam.p, P->nb_cam,
                       (const double*)P->part_cost.p, P->nb_cost, (int)SC_N, P->mail_dev, P->mail_count.p, inexact ? (int)SC_ZL8 : -1);
    P->mail_expected += 1.0;
    return mail_wait(P, h, SC_N * sizeof(double));
  }
  launch_step(P, inexact);
  if (int st = launch_cost(P, P->q_trial.p, SC_TRIAL)) return st;
  return read_scalars(P, h);
}

// Per-camera host arrays (rotations, gradient, mat-vec operands) enter and leave in the caller's numbering.
const double* to_internal(gsfm_rot_problem* P, const double* ext, int width) {
  if (P->perm.empty()) return ext;
  P->h_cam.resize((size_t)P->n_cams * width);
  for (size_t k = 0; k < P->n_cams; ++k) std::memcpy(&P->h_cam[(size_t)P->perm[k] * width], ext + k * width, 8 * (size_t)width);
  return P->h_cam.data();
}
void to_external(gsfm_rot_problem* P, const double* internal, double* ext, int width) {
  for (size_t k = 0; k < P->n_cams; ++k) std::memcpy(ext + k * width, internal + (size_t)P->perm[k] * width, 8 * (size_t)width);
}

int upload_state(gsfm_rot_problem* P, const double* rot_aa) {
  const size_t N = P->n_cams;
  HIPCHK(hipMemcpyAsync(P->aa_io.p, to_internal(P, rot_aa, 3), 24 * N, hipMemcpyHostToDevice, P->stream));
  if (P->param_dim == 3) { HIPCHK(hipMemcpyAsync(P->x.p, P->aa_io.p, 24 * N, hipMemcpyDeviceToDevice, P->stream)); }
  else {  // estimator.cpp:130-136: angle-axis -> quaternion state
    hipLaunchKernelGGL(k_cam_cache, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->aa_io.p, P->n_cams, 3, (double2*)P->x.p);
  }
  launch_cache(P, P->x.p, P->q.p);
  return 0;
}
// The same two transfers for a caller whose rotations live in device memory (gsfm_rot_solve_resident): nothing crosses PCIe.
int upload_state_resident(gsfm_rot_problem* P, const double* d_rot_aa) {
  const size_t N = P->n_cams;
  if (P->perm.empty()) { HIPCHK(hipMemcpyAsync(P->aa_io.p, d_rot_aa, 24 * N, hipMemcpyDeviceToDevice, P->stream)); }
  else {
    if (!P->d_perm.p && P->d_perm.upload(P->perm) != hipSuccess) { (void)hipGetLastError(); return fail(GSFM_ERR_HIP, "uploading the camera relabelling failed"); }
    hipLaunchKernelGGL(k_cam_permute3, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, d_rot_aa, (const uint32_t*)P->d_perm.p, P->n_cams, 1, P->aa_io.p);
  }
  if (P->param_dim == 3) { HIPCHK(hipMemcpyAsync(P->x.p, P->aa_io.p, 24 * N, hipMemcpyDeviceToDevice, P->stream)); }
  else hipLaunchKernelGGL(k_cam_cache, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->aa_io.p, P->n_cams, 3, (double2*)P->x.p);
  launch_cache(P, P->x.p, P->q.p);
  return 0;
}
int download_state_resident(gsfm_rot_problem* P, double* d_rot_aa) {
  const size_t N = P->n_cams;
  const double* src = P->x.p;
  if (P->param_dim != 3) {
    hipLaunchKernelGGL(k_quat_to_aa, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->active.p, P->n_cams, P->aa_io.p);
    src = P->aa_io.p;
  }
  if (P->perm.empty()) { HIPCHK(hipMemcpyAsync(d_rot_aa, src, 24 * N, hipMemcpyDeviceToDevice, P->stream)); }
  else hipLaunchKernelGGL(k_cam_permute3, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, src, (const uint32_t*)P->d_perm.p, P->n_cams, 0, d_rot_aa);
  return sync_check(P, "rotations (device-resident)");
}
int download_state(gsfm_rot_problem* P, double* rot_aa) {
  const size_t N = P->n_cams;
  double* dst = rot_aa;
  if (!P->perm.empty()) { P->h_cam.resize(3 * N); dst = P->h_cam.data(); }
  if (P->param_dim == 3) { HIPCHK(hipMemcpyAsync(dst, P->x.p, 24 * N, hipMemcpyDeviceToHost, P->stream)); }
  else {
    hipLaunchKernelGGL(k_quat_to_aa, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->active.p, P->n_cams, P->aa_io.p);
    HIPCHK(hipMemcpyAsync(dst, P->aa_io.p, 24 * N, hipMemcpyDeviceToHost, P->stream));
  }
  if (int st = sync_check(P, "download rotations")) return st;
  if (!P->perm.empty()) to_external(P, dst, rot_aa, 3);
  return 0;
}

// ---- TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy (ceres 1.14 semantics) ----
int lm_solve(gsfm_rot_problem* P, const gsfm_rot_options& o_in, gsfm_rot_summary* sum) {
  // Several scenes batched as one disconnected graph (BASELINE C4): the PCG stopping rule is a GLOBAL relative residual, so a
  // component whose gradient is already orders of magnitude below the others' is allowed an error that is large against its own
  // right-hand side, and at vanishing damping that error lands in its weakly determined directions.  Measured on the 14-scene batch
  // with the real Madrid graph inside: 3e-5 rad on Madrid's cameras at 1e-12, 4e-9 at 1e-14 (for 9 % more PCG iterations; the
  // reference's Cholesky solves every block exactly).  Disconnected problems therefore never run looser than 1e-14.
  gsfm_rot_options o = o_in;
  if (P->n_components > 1) o.cg_relative_tolerance = std::min(o.cg_relative_tolerance, 1e-14);
  if (o.verbose && o.cg_relative_tolerance != o_in.cg_relative_tolerance)
    fprintf(stderr, "[gsfm] the view graph has %u connected components: PCG runs to a relative residual of %.0e instead of the requested %.0e\n", P->n_components, o.cg_relative_tolerance, o_in.cg_relative_tolerance);
  const double t0 = now_ms();
  std::memset(sum, 0, sizeof(*sum));
  sum->iters_to_1e6 = -1;
  sum->num_edges_used = P->cost.n;
  P->trace.clear();
  P->timer.acc[0] = P->timer.acc[1] = P->timer.acc[2] = 0;
  P->graph_launches = 0;
  P->n_collectives = P->n_pcg_collectives = P->n_pcg_launched = 0;
  P->lap = P->lap_capable;
  P->comps.fresh_solve = true;
  P->component_rest = o.component_rest != 0;
  double h[SC_N];
  double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
  int num_invalid = 0, iteration = 0;
  double x_cost = 0, x_norm = 0, gmax = 0;

  auto record = [&](double cost, double dc, double sn, double rd, int cg) {
    const double row[GSFM_ROT_TRACE_COLS] = {(double)iteration, cost, dc, gmax, sn, rd, radius, (double)cg};
    P->trace.insert(P->trace.end(), row, row + GSFM_ROT_TRACE_COLS);
    if (o.verbose) fprintf(stderr, "[gsfm] it %3d cost %.12e dcost %.3e |g| %.3e |dx| %.3e rho %.3e radius %.3e cg %d\n",
                           iteration, cost, dc, gmax, sn, rd, radius, cg);
  };
  bool spec_enqueued = false, exact_pipeline_used = false;   // LM control on the device: an exact iteration is already in flight / the pipeline ran at all
  // Host-controlled steps: the gradient's max norm of an accepted point is not waited for.  Its only consumer is the gradient-tolerance test at
  // the top of the NEXT iteration, so it is copied to pinned memory behind the damping rebuild and read at that iteration's first host
  // synchronisation (the PCG's first look, or the trial cost's) -- one read-back + idle GPU (25-35 us) fewer per accepted step; if the test then
  // fires, the linear solve that was started is discarded (nothing of it had been applied) and the run ends where it would have.  The trace row of
  // the accepting iteration gets its |g| when it arrives.  (Verbose runs: the round-3 read-back, so that every line is complete when printed.)
  const bool defer_gmax = P->pin && !o.verbose;
  bool gmax_deferred = false;
  size_t gmax_trace_slot = 0;
  volatile double* const gmax_pin = P->pin ? (volatile double*)((char*)P->pin + 256) : nullptr;
  auto take_gmax = [&]() {   // (call behind a synchronisation of the stream)
    gmax = *gmax_pin; gmax_deferred = false;
    if (gmax_trace_slot < P->trace.size()) P->trace[gmax_trace_slot] = gmax;
  };
  auto finish = [&](int term) {
    if (gmax_deferred || P->timer.used) { (void)hipStreamSynchronize(P->stream); P->timer.resolve(); if (gmax_deferred) take_gmax(); }   // (the mailbox reads leave the phase timers' events unresolved)
    if (spec_enqueued || exact_pipeline_used) { (void)hipStreamSynchronize(P->stream); P->timer.resolve(); spec_enqueued = false; }   // (whatever was enqueued ahead skips itself; the phase timers need the sync)
    sum->termination = term; sum->num_iterations = iteration; sum->final_cost = x_cost; sum->final_gradient_max_norm = gmax;
    sum->final_radius = radius; sum->t_total_ms = now_ms() - t0;
    sum->num_graph_launches = P->graph_launches;
    sum->num_collectives = P->n_collectives; sum->num_pcg_collectives = P->n_pcg_collectives; sum->num_pcg_launched = P->n_pcg_launched;
    sum->t_linearize_ms = P->timer.acc[T_LIN]; sum->t_sweep_ms = P->timer.acc[T_SWEEP]; sum->t_cg_ms = P->timer.acc[T_CG];
    if (!std::isfinite(x_cost)) sum->nonfinite = 1;
    return 0;
  };

  // Init + IterationZero
  hipLaunchKernelGGL(k_cam_norm, dim3(P->nb_cam), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->active.p, P->n_cams, P->param_dim, P->part_cam.p);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cam.p, P->nb_cam, P->scal.p + SC_XNORM2);
  if (int st = launch_cost(P, P->q.p, SC_COST)) return st;
  if (int st = launch_lin(P, P->q.p)) return st;
  sum->num_residual_sweeps++; sum->num_linearizations++;
  launch_prep(P, o, radius, true);
  bool prep_valid = true;
  if (int st = read_scalars(P, h)) return st;
  x_cost = h[SC_COST]; gmax = h[SC_GMAX]; x_norm = std::sqrt(h[SC_XNORM2]);
  sum->initial_cost = x_cost;
  record(x_cost, 0, 0, 0, 0);
  if (!std::isfinite(x_cost)) return finish(GSFM_TERM_FAILURE);
  if (gmax <= o.gradient_tolerance) return finish(GSFM_TERM_GRADIENT_TOLERANCE);
  bool last_successful = false, pcg_struggles = false, pcg_dearer_than_cholesky = false;
  // What one exact step (assemble + factorise + both substitutions, one hipGraph replay) takes on MI355X as a function of 3N: measured
  // (tools/bench_chol_large.hip, profiles/r06b_chol_look.txt: 1182 -> 0.40 ms, 2400 -> 0.89, 4500 -> 2.25, 9000 -> 11.8 with the schedule
  // of round 6; rounds 3-5: 0.45 / 1.33 / 3.27 / 13.5), log-log interpolated.
  auto dense_cost_ms = [](double n3) {
    static const double pts[5][2] = {{600.0, 0.18}, {1182.0, 0.40}, {2400.0, 0.89}, {4500.0, 2.25}, {9000.0, 11.8}};
    int k = 0;
    while (k < 3 && n3 > pts[k + 1][0]) ++k;
    const double t = std::log(n3 / pts[k][0]) / std::log(pts[k + 1][0] / pts[k][0]);
    return pts[k][1] * std::pow(pts[k + 1][1] / pts[k][1], t);
  };
  // LM control on the device for exact steps (GSFM_LM_DEVICE_CONTROL=0: the host loop, for A/B and for the bit-identity test): unsharded
  // problems with a native loss on the row-major layout -- the linearisation of the accept path must be enqueueable without the host

  bool exact_pipeline_broken = false;   // exact steps turned out impossible (size, memory)
  const bool device_control = o.lm_device_control != 0 && !P->sharded && !P->cb && !P->cs.active;
  // Forcing schedule (gsfm_rot_options::pcg_forcing): steps far from convergence may deviate from the exact step by at most `eps_rad` (rms over the
  // cameras); off for disconnected graphs (their 1e-14 rule stands).
  const double eps_rad = o.pcg_forcing_tolerance, tau_max = 1e-2, sqrt_n = std::sqrt((double)std::max<uint32_t>(1, P->n_cams));
  // ... and no step is asked for a relative energy error below kappa * |step|_rms (kappa = 5e-6 per radian): a Gauss-Newton step is itself
  // only accurate to O(|step|^2) -- the linearisation error, which the following iterations correct -- so for steps of several degrees, far
  // from convergence, a linear solve to 1e-8 rad would be wasted on it; the deviation allowed, kappa |step|^2, stays orders below that error
  // (a 14-degree step: tau 1.2e-6 instead of 4e-8; below 2.6 degrees the absolute bound is the tighter one).  Tree start of the benchmark
  // graph: 683 -> ~500 PCG iterations, the final answer ~1e-9 rad (mean) from the exact schedule (profiles/r04_forcing_floor.txt).
  // Larger floors were swept at the end of the round (profiles/r04b_kappa_sweep.txt: 5e-5 / 1e-4 / 3e-4 -> tree start 437 / 378 / 350 iterations,
  // 97 / 85 / 80 ms, C5 36 / 35 / 34 iterations, the same two misses among 40 random graphs up to 1e-4) and NOT adopted: at 1e-4 the suite's own
  // forcing pass loses a 172-iteration trajectory (trust region creeping up, steps above 0.6 degrees for dozens of iterations: the deviations of
  // consecutive steps add up faster than the slow convergence contracts them -- 5e-4 rad from the oracle, where 5e-6 follows it).
  constexpr double kappa = 5e-6;   // (swept again under the contraction gate in round 5: 1e-4 / 3e-4 buy the tree start 7 % and lose a 13-iteration MAGSAC trajectory, profiles/r05_kappa_sweep.txt)
  // (not for QUATERNION_NORM: that functor canonicalises the sign of two quaternions separately, quat.hpp:135-142 -- a DISCONTINUOUS residual, where a
  // 1e-8 rad difference in an iterate flips signs the exact schedule does not flip; tests/manual/fuzz_forcing.py found it)
  // (and not for disconnected graphs: tried on C4 -- the 14-scene batch ended after 30 LM iterations instead of the oracle's 46, 36.8 instead of 86.4 ms: a
  // component that has nearly converged while the batch iterates on gets the share of a loose solve its share of the energy asks for, i.e. none, its
  // cost change vanishes and the global function-tolerance test fires early; the energy norm of the whole step says nothing about one component)
  // (and not under a loss that switches edges OFF -- Tukey: rho' is identically zero beyond a^2 -- or one the library cannot see into, a host callback:
  // a camera whose edges are all but cut off has next to no weight of its own, so no norm a loose solve can be stopped on -- energy, residual,
  // block-Jacobi's per-camera estimate -- bounds how far it is left from its exact step, and it is exactly those cameras whose next
  // linearisation decides which of their edges come back.  tests/manual/fuzz_forcing.py, dense graphs 5:14 / 6:52: 3 to 5 LM iterations, every
  // step contracting, 8e-7 / 3.6e-6 rad mean and 1e-4 max on a handful of cameras at every per-step tolerance down to 1e-9 rad.  pcg_forcing = 3 forces it on.)
  const bool forcing = o.pcg_forcing > 0 && P->n_components <= 1 && eps_rad > 0.0 && P->functor != F_QNORM && ((!P->loss_cuts_off && !P->cb) || o.pcg_forcing == 3);
  double pred_rms = -1.0;          // rms size of the last accepted step: the (conservative: steps shrink) prediction of the next one's
  // Contraction gate of the forcing schedule (round 5).  An inexact step leaves the iterate ~eps_rad away from the reference's trajectory; whether
  // that matters at the END is a property of the trajectory: where consecutive steps shrink fast (every benchmark configuration from a sensible
  // start: |step_k| / |step_k-1| = 0.06-0.21) the following steps forget it; on slow trajectories -- far starts under redescending or cut-off
  // losses: ratios 0.3-0.95 for dozens of iterations, each ending by the function tolerance a step or more away from the minimum -- they do
  // not, and under the MAGSAC losses (rho and rho' piecewise constant in s: the table index is ROUNDED, scripts/loss_functions.py:285-341)
  // ANY deviation that moves one edge across a table cell puts the run on a different self-consistent weighting, a fixed distance away
  // whatever the size of the deviation (profiles/r05_forcing_probe.txt: the same 5e-6 / 1e-5 rad at every PCG tolerance from 1e-8 to 1e-3,
  // 1e-14 at 1e-9).  So: loose solves only while every accepted step so far was at most `contraction_max` times its predecessor; the first
  // violation switches the schedule off for the rest of the run, and if an inexact step has already been applied the run is REDONE from the
  // initial rotations with exact steps (GSFM_INTERNAL_RESTART; typically after two cheap loose steps of a run that needs dozens).  A run that
  // completes under the schedule therefore carries the deviation of its last inexact step plus a geometric tail of the earlier ones.
  constexpr double contraction_max = 0.3;
  bool forcing_live = forcing, loose_applied = false;
  double prev_accepted_norm = -1.0;
  while (true) {
    if (iteration >= o.max_num_iterations) return finish(GSFM_TERM_NO_CONVERGENCE);
    if (last_successful && !gmax_deferred && gmax <= o.gradient_tolerance) return finish(GSFM_TERM_GRADIENT_TOLERANCE);
    if (radius <= o.min_trust_region_radius) {
      if (gmax_deferred) { if (int st = sync_check(P, "gradient norm")) return st; take_gmax(); if (gmax <= o.gradient_tolerance) return finish(GSFM_TERM_GRADIENT_TOLERANCE); }
      return finish(GSFM_TERM_FAILURE);
    }
    ++iteration;
    last_successful = false;
    if (!prep_valid) launch_prep(P, o, radius, false);
    prep_valid = false;
    int cg = 0, cg_spent = 0; double cg_rel = 0;
    bool dense_used = false;
    // Forcing schedule: the step is solved loosely -- to a relative (energy-norm) error tau chosen so that tau * |step|_rms <= eps_rad, with the
    // step size predicted from the previous accepted step (first step: tau_max, corrected below) -- unless it is the last one the iteration
    // cap allows (that one is applied whatever it looks like: exact).
    bool loose = forcing_live && iteration < o.max_num_iterations;
    double tau = pred_rms > 0.0 ? std::fmin(tau_max, std::fmax(kappa * pred_rms, eps_rad / pred_rms)) : tau_max;
    if (tau <= 4.0 * o.cg_relative_tolerance) loose = false;
    bool use_pcg2 = false;
    // dense_cholesky_max_cams > 0: exact Cholesky steps for graphs up to that size; < 0: up to |value| cameras, but only
    // once a PCG solve of this run has needed more than 150 iterations (2.5 ms of factorisation beats that many mat-vecs)
    // dense_cholesky_auto_cams: graphs beyond dense_cholesky_max_cams and up to this size switch to exact steps from the moment a PCG-solved
    // step has cost more GPU time than a factorisation of their size is known to take (dense_cost_ms below): sticky for the rest of the solve.
    const int64_t dense_cap = o.dense_cholesky_max_cams > 0 ? o.dense_cholesky_max_cams : -(int64_t)o.dense_cholesky_max_cams;
    const bool dense_auto = o.dense_cholesky_max_cams > 0 && (int64_t)P->n_cams > dense_cap && (int64_t)P->n_cams <= (int64_t)o.dense_cholesky_auto_cams && pcg_dearer_than_cholesky;
    const bool exact_now = !P->sharded && ((dense_cap > 0 && (int64_t)P->n_cams <= dense_cap && (o.dense_cholesky_max_cams > 0 || pcg_struggles)) || dense_auto);
    if (gmax_deferred && exact_now) {   // (exact steps do not pass a synchronisation before they need it)
      if (int st = sync_check(P, "gradient norm")) return st;
      take_gmax();
      if (gmax <= o.gradient_tolerance) { --iteration; return finish(GSFM_TERM_GRADIENT_TOLERANCE); }
    }
    if (exact_now && device_control && !exact_pipeline_broken) {
      // Exact step with the trust-region decisions on the device (kernels.hpp, k_lm_decide): the whole LM iteration -- factorisation, step,
      // trial cost, decision, predicated accept path, damping for the next step -- is enqueued without a host decision, and iteration k + 1
      // is enqueued BEFORE iteration k's record is read (from a side stream), so the GPU never waits for the host between two exact steps.
      // An iteration enqueued ahead of a verdict that ends the run (termination, broken factor) decides nothing (CT_SKIPPED).
      double* ctl = P->scal.p + SC_CTL;
      const LmOpts lo{o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance, o.min_relative_decrease, o.max_trust_region_radius, o.min_trust_region_radius};
      constexpr int REC = CT_N + 1;
      if (!P->rec_host) {
        if (hipHostMalloc((void**)&P->rec_host, 4 * REC * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void**)&P->rec_dev, P->rec_host, 0) != hipSuccess) { (void)hipGetLastError(); return fail(GSFM_ERR_HIP, "mapped record of the LM control"); }
      }
      // (no phase timers in this pipeline: every begin / end is an event record on the stream, eight of them per iteration = 24 us of the
      // 545 us a Madrid iteration takes -- 34.3 -> 32.6-33.0 ms per solve; the summary's t_*_ms cover the host-controlled steps only)
      struct Mute { EventTimer& t; explicit Mute(EventTimer& tt) : t(tt) { t.mute = true; } ~Mute() { t.mute = false; } };
      double* const it_dev = P->scal.p + SC_REC;   // the device-resident iteration number (k_lm_set / k_lm_after)
      auto enqueue_kernels = [&](bool capturing) -> int {   // 0: enqueued, 1: no exact step possible (size, memory), < 0: error
        bool used = false;
        if (int st = run_dense(P, &used, capturing)) return -st;
        if (!used) return 1;
        // (the single-workgroup reductions ride in their consumers: k_lm_decide sums the step's and the trial cost's partials and applies an
        // accepted step, k_lm_after takes the gradient's max norm -- six launches after the factorisation instead of ten)
        launch_step(P, false, false);
        if (int st = launch_cost(P, P->q_trial.p, SC_TRIAL, CostOutputs(), false)) return -st;
        hipLaunchKernelGGL(k_lm_decide, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, lo, P->scal.p, (int)SC_STEP, (int)SC_TRIAL, (int)SC_DENSE_INFO, ctl,
                           (const double*)P->part_cam.p, P->nb_cam, (const double*)P->part_cost.p, P->nb_cost,
                           P->n_cams, P->param_dim, P->x.p, (const double*)P->x_trial.p, P->q.p, (const double2*)P->q_trial.p);
        if (int st = launch_lin(P, P->q.p, ctl + CT_ACCEPT)) return -st;
        launch_prep(P, o, radius, false, ctl + CT_RADIUS, false);
        hipLaunchKernelGGL(k_lm_after, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, lo, P->scal.p, (int)SC_GMAX, ctl, P->rec_dev, (int)REC, it_dev, (const double*)P->part_cam.p, P->nb_cam);
        return 0;
      };
      // (The whole iteration as ONE hipGraph -- asked for by three reviews, built in round 4 -- measured neutral on Madrid, 32.48 against 32.45 ms:
      // the factorisation already replays as a graph and the six launches behind it are queued while it runs.  profiles/r04b_iter_graph_ab.txt; removed in round 5.)
      auto enqueue_exact = [&](int it) -> int {
        const Mute muted(P->timer);
        P->rec_host[REC * (it & 3) + CT_N] = -1.0;   // (the slot's previous user, iteration it - 4, was read long ago)
        return enqueue_kernels(false);
      };
      int eq = 0;
      exact_pipeline_used = true;
      if (!spec_enqueued) {
        hipLaunchKernelGGL(k_lm_set, dim3(1), dim3(1), 0, P->stream, ctl, radius, decrease_factor, x_cost, x_norm, gmax, (double)num_invalid, it_dev, (double)iteration);
        eq = enqueue_exact(iteration);
        if (eq < 0) return -eq;
      }
      if (eq == 1) exact_pipeline_broken = true;   // (falls through to the generic path below: PCG)
      else {
        spec_enqueued = false;
        if (iteration + 1 <= o.max_num_iterations) {
          const int e2 = enqueue_exact(iteration + 1);
          if (e2 < 0) return -e2;
          spec_enqueued = e2 == 0;
        }
        double c[CT_N];
        {   // this iteration's record: poll its stamp in mapped host memory (the main stream may already be running the next iteration)
          volatile double* slot = P->rec_host + REC * (iteration & 3);
          const double t_poll = now_ms();
          bool seen = false;
          for (long spin = 0; !(seen = slot[CT_N] == (double)iteration); ++spin) {
            if ((spin & 0x3ff) == 0x3ff && now_ms() - t_poll > 20000.0) break;   // (20 s: something is badly wrong)
            __builtin_ia32_pause();
          }
          if (!seen) {   // fall back to the stream: an error on it surfaces here
            if (int st = sync_check(P, "LM control: record")) return st;
            if (slot[CT_N] != (double)iteration) return fail(GSFM_ERR_HIP, "LM control: the record of an iteration never arrived");
          }
          std::atomic_thread_fence(std::memory_order_acquire);
          for (int k = 0; k < CT_N; ++k) c[k] = slot[k];
        }
        prep_valid = true;                                 // (rebuilt on the device with the radius decided there)
        if (c[CT_DENSE_FAIL] != 0.0) {
          // the factorisation met a non-positive pivot: nothing was decided, the linearisation did not run, the damping was rebuilt from the
          // unchanged radius, whatever was enqueued ahead is skipping itself; PCG solves this step again under host control
          if (int st = sync_check(P, "LM control: drain")) return st;
          spec_enqueued = false;
        } else {
          sum->num_dense_solves++;
          sum->num_residual_sweeps++;
          num_invalid = (int)c[CT_NINVALID];
          auto leave = [&](int term) { return finish(term); };
          if (c[CT_VALID] == 0.0) {                          // HandleInvalidStep
            if (c[CT_TERM] == 4.0) return leave(GSFM_TERM_FAILURE);
            radius = c[CT_RADIUS]; decrease_factor = c[CT_DF];
            sum->num_unsuccessful_steps++;
            record(x_cost, 0, 0, 0, 0);
            continue;
          }
          if (c[CT_NONFINITE] != 0.0) sum->nonfinite = 1;
          const double step_norm = c[CT_STEPN], cost_change = c[CT_CC], rel_dec = c[CT_CC] / c[CT_MCC];
          if (sum->iters_to_1e6 < 0 && std::fabs(cost_change) <= 1e-6 * x_cost) sum->iters_to_1e6 = iteration;
          if (c[CT_TERM] == 2.0) { record(x_cost, cost_change, step_norm, rel_dec, 0); return leave(GSFM_TERM_PARAMETER_TOLERANCE); }
          if (c[CT_TERM] == 0.0) { record(x_cost, cost_change, step_norm, rel_dec, 0); return leave(GSFM_TERM_FUNCTION_TOLERANCE); }
          radius = c[CT_RADIUS]; decrease_factor = c[CT_DF];
          if (c[CT_ACCEPT] != 0.0) {
            x_norm = c[CT_XNORM]; x_cost = c[CT_XCOST]; gmax = c[CT_GMAX];
            sum->num_residual_sweeps++; sum->num_linearizations++;
            sum->num_successful_steps++;
            last_successful = true;
            pred_rms = step_norm / sqrt_n;
            // (the contraction gate, as behind a host-controlled step: exact steps that follow inexact ones -- a run the factorisation took over -- are held to it too)
            if (forcing_live && prev_accepted_norm > 0.0 && step_norm > contraction_max * prev_accepted_norm) {
              forcing_live = false;
              if (loose_applied) { record(x_cost, cost_change, step_norm, rel_dec, 0); finish(GSFM_TERM_NO_CONVERGENCE); return GSFM_INTERNAL_RESTART; }
            }
            prev_accepted_norm = step_norm;
          } else sum->num_unsuccessful_steps++;
          record(x_cost, cost_change, step_norm, rel_dec, 0);
          continue;
        }
      }
    }
    if (exact_now && !exact_pipeline_broken && !(device_control)) {
      if (int st = run_dense(P, &dense_used)) return st;
    }
    // A TIGHT solve that ends above its tolerance (iteration cap, stagnation: run_pcg) is not the reference's exact step (estimator.cpp:300).  Where
    // the factorisation exists for the size it takes over for this step and the rest of the run; elsewhere the step is evaluated all the same
    // and COUNTED (gsfm_rot_summary::num_pcg_capped_steps, worst_accepted_cg_residual) -- never silently.
    const bool dense_rescue = !P->sharded && o.dense_cholesky_max_cams > 0 && (int64_t)P->n_cams <= (int64_t)o.dense_cholesky_auto_cams;
    bool dense_failed = false;
    // Disconnected view graph: the small components factorised exactly, side by side, PCG on the large ones (solver_components.hpp)
    bool comp_used = false;
    if (P->packed) (void)hipMemsetAsync(P->scal.p + SC_COMPBAD, 0, sizeof(double), P->stream);
    if (!dense_used && P->n_components > 1 && (!P->sharded || P->packed)) {
      if (int st = run_component_step(P, o, o_in.cg_relative_tolerance, pcg_struggles, radius >= o.initial_trust_region_radius, &comp_used, &cg, &cg_rel)) return st;
      if (comp_used) {
        loose = false;
        if (gmax_deferred) {
          if (P->comps.all_dense) { if (int st = sync_check(P, "gradient norm")) return st; }
          take_gmax();
          if (gmax <= o.gradient_tolerance) { --iteration; return finish(GSFM_TERM_GRADIENT_TOLERANCE); }
        }
        if (P->packed) { if (int st = packed_exchange(P)) return st; }
        if (int st = evaluate_trial(P, false, h)) return st;
        int info = 0;
        std::memcpy(&info, &h[SC_DENSE_INFO], sizeof(int));
        if (P->packed) info = h[SC_COMPBAD] != 0.0;   // (any rank's: all of them take the fallback together)
        if (info != 0) { comp_used = false; cg_spent += cg; cg = 0; if (P->packed) (void)hipMemsetAsync(P->scal.p + SC_COMPBAD, 0, sizeof(double), P->stream); }   // a component's factor broke down: plain PCG solves the whole step
        else { sum->num_dense_solves++; dense_used = P->comps.all_dense; }
      }
    }
    for (int attempt = 0; attempt < 2 && !comp_used; ++attempt) {
      if (!dense_used) {
        if (int st = coarse_build(P, pcg_struggles)) return st;
        use_pcg2 = P->coarse_n == 0 && use_single_reduction(P, o);
        if (int st = (use_pcg2 ? run_pcg2(P, o, o.cg_relative_tolerance, loose ? tau * tau : 0.0, -1, &cg, &cg_rel) : run_pcg(P, o, o.cg_relative_tolerance, loose ? tau * tau : 0.0, -1, &cg, &cg_rel))) return st;
        if (gmax_deferred) {
          take_gmax();
          if (gmax <= o.gradient_tolerance) { --iteration; return finish(GSFM_TERM_GRADIENT_TOLERANCE); }   // (this iteration never began)
        }
        if (!loose && cg_rel > o.cg_relative_tolerance && dense_rescue && !dense_failed && !exact_pipeline_broken) {
          if (int st = run_dense(P, &dense_used)) return st;
          if (dense_used) pcg_dearer_than_cholesky = true;
        }
      }
      if (P->packed && !dense_used) { if (int st = packed_exchange(P)) return st; }
      if (int st = evaluate_trial(P, !dense_used && loose, h)) return st;
      if (P->packed && attempt == 0 && h[SC_COMPBAD] != 0.0) {   // another rank's component factorisation broke down: it solves this step again by PCG, and so does everybody (one more exchange on every rank)
        (void)hipMemsetAsync(P->scal.p + SC_COMPBAD, 0, sizeof(double), P->stream);
        cg_spent += cg; cg = 0;
        continue;
      }
      for (int pass = 0; pass < 4 && !dense_used && loose && cg_rel > o.cg_relative_tolerance; ++pass) {
        // The loose step has been evaluated.  Every decision the trust-region loop takes from it must be the one the exact step would give:
        //  * termination (function / parameter tolerance): the decisive quantities -- cost change, step norm -- of the loose step are within
        //    O(tau) of the exact step's (the model decrease even within O(tau^2)), so a value more than a factor two away from its threshold
        //    decides (a terminating step is never applied: the answer is the same); inside that band PCG continues to the tight tolerance;
        //  * acceptance: a relative decrease above 0.25 (the threshold is 1e-3) decides, anything else is settled on the exact step, as is an
        //    invalid model;
        //  * an accepted loose step must also be within eps_rad (rms, estimated: tau * |step|_rms) of the exact one now that its size is known,
        //    otherwise PCG continues to the tau that size asks for.
        // "Continues": from the state the stop left, i.e. the same iterates as an uninterrupted solve at the new tolerance.
        const double mcc = -0.5 * h[SC_STEP] + 0.5 * h[SC_STEP + 1] + 0.5 * h[SC_STEP + 2];
        const double cc = x_cost - h[SC_TRIAL], sn = std::sqrt(h[SC_STEP + 3]);
        const double pt = o.parameter_tolerance * (x_norm + o.parameter_tolerance), ft = o.function_tolerance * x_cost;
        const bool valid_l = std::isfinite(mcc) && mcc > 0.0 && std::isfinite(h[SC_TRIAL]);
        bool tight = !valid_l || o.pcg_forcing == 2;
        // Conditioning gate (staircase losses): a loose solve that has needed more than 64 iterations says the preconditioned system is ill
        // conditioned -- tight solves of such runs take hundreds -- and there the energy estimate says little about the weakly coupled camera
        // clusters whose mode emerges last.  Under a smooth loss what they are left short of is made up by the next steps; under MAGSAC it decides
        // which of their edges enter the inlier band (fuzz seed 2 trial 45: 553 / 84 / 11 / 9 loose against 1 253 / 295 / 146 / 176 tight
        // iterations, every step contracting, a handful of cameras 1.7e-2 rad elsewhere).  The solve continues to the tight tolerance and the
        // schedule is off for the rest of the run; a run that had applied an inexact step before is redone.
        if (!tight && P->loss_staircase && cg > 64 && o.pcg_forcing != 3) {
          tight = true; forcing_live = false;
          if (loose_applied) { finish(GSFM_TERM_NO_CONVERGENCE); return GSFM_INTERNAL_RESTART; }
        }
        double tau_need = tau;
        if (!tight) {
          if (sn <= 0.5 * pt || std::fabs(cc) <= 0.5 * ft) { sum->num_inexact_steps++; break; }            // terminates, as the exact step would
          if (sn <= 2.0 * pt || std::fabs(cc) <= 2.0 * ft || cc / mcc <= std::fmax(o.min_relative_decrease, 0.25)) tight = true;
          else {
            tau_need = std::fmax(kappa * sn / sqrt_n, eps_rad / std::fmax(sn / sqrt_n, 1e-300));
            // ... and no single camera may be left far from its exact step: block-Jacobi's estimate of what each camera's step still lacks
            // (k_cam_step's sixth sum, a smooth maximum in radians) is held against 10 x the rms tolerance -- the energy norm does not see a
            // camera whose weights have all but vanished (kernels.hpp, StepArgs::Minv)
            const double zmax = std::pow(std::fmax(h[SC_ZL8], 0.0), 0.125) * (P->param_dim == 4 ? 2.0 : 1.0);
            constexpr double zcap = 100.0;   // (10 and "off" gave the same 420 fuzz outcomes; 10 cost C5 13 iterations: profiles/r05_fuzz_forcing.txt)
            const bool cams_ok = zmax <= zcap * eps_rad;
            if (!cams_ok) tau_need = std::fmin(tau_need, 0.5 * tau * (zcap * eps_rad / zmax));
            if (cams_ok && tau <= 1.5 * tau_need) { sum->num_inexact_steps++; break; }
            if (tau_need <= 4.0 * o.cg_relative_tolerance || pass == 3) tight = true;
          }
        }
        if (tight) { loose = false; tau = 0.0; } else tau = std::fmin(tau, tau_need);
        if (int st = (use_pcg2 ? run_pcg2(P, o, o.cg_relative_tolerance, tau * tau, cg, &cg, &cg_rel) : run_pcg(P, o, o.cg_relative_tolerance, tau * tau, cg, &cg, &cg_rel))) return st;
        if (!loose && cg_rel > o.cg_relative_tolerance && dense_rescue && !dense_failed && !exact_pipeline_broken) {   // (the continued solve ran into the cap)
          if (int st = run_dense(P, &dense_used)) return st;
          if (dense_used) pcg_dearer_than_cholesky = true;
        }
        if (int st = evaluate_trial(P, !dense_used && loose, h)) return st;
        sum->num_forcing_refinements++;
        if (!loose) break;
      }
      if (!dense_used) break;
      int info = 0;
      std::memcpy(&info, &h[SC_DENSE_INFO], sizeof(int));
      if (info == 0) { sum->num_dense_solves++; break; }
      dense_used = false; dense_failed = true; cg_spent += cg; cg = 0;   // not positive definite to working precision: the step just evaluated is meaningless, PCG solves it again
    }
    if (!dense_used && !P->sharded && (int64_t)P->n_cams <= (int64_t)o.dense_cholesky_auto_cams) {
      // What this step's linear solve cost against a factorisation of its size -- from its ITERATION COUNT and a cost per iteration (dependent
      // launches: ~8 us for the single-reduction recurrence, ~14 us for the textbook one, measured at these sizes, + the mat-vec's stream), not
      // from a clock: the choice is the same on every run and every box (round-4 advisor).
      const double pcg_model_ms = cg * ((use_pcg2 ? 8e-3 : 14e-3) + 52.0 * (double)P->dir.n / 4e9);
      if (pcg_model_ms > 1.25 * dense_cost_ms(3.0 * P->n_cams)) pcg_dearer_than_cholesky = true;
    }
    const bool step_loose = !dense_used && loose && cg_rel > o.cg_relative_tolerance;
    if (!dense_used && !step_loose) {
      sum->worst_accepted_cg_residual = std::fmax(sum->worst_accepted_cg_residual, cg_rel);
      if (cg_rel > o.cg_relative_tolerance) {
        sum->num_pcg_capped_steps++;
        if (o.verbose) fprintf(stderr, "[gsfm] it %3d: PCG stopped after %d iterations with a relative residual of %.1e (tolerance %.1e): this step is inexact\n", iteration, cg, cg_rel, o.cg_relative_tolerance);
      }
    }
    // (a loose solve's count is projected to the tight tolerance -- PCG converges about linearly in the logarithm -- before it is held against the 150)
    if ((loose && tau > 0.0 ? cg * std::log(o.cg_relative_tolerance) / std::log(std::fmin(0.5, tau)) : (double)cg) > 150.0) pcg_struggles = true;
    sum->num_cg_iterations += cg_spent + cg;   // (a solve the factorisation took over: its PCG iterations were spent all the same)
    sum->num_residual_sweeps++;
    // model_cost_change = -eta.g - 1/2 eta^T B eta with B eta = -g - r_cg - Lambda eta
    const double eta_g = h[SC_STEP], eta_r = h[SC_STEP + 1], eta_L = h[SC_STEP + 2];
    const double model_cost_change = -0.5 * eta_g + 0.5 * eta_r + 0.5 * eta_L;
    const bool valid = std::isfinite(model_cost_change) && model_cost_change > 0.0;
    if (!valid) {  // HandleInvalidStep
      if (++num_invalid >= 5) return finish(GSFM_TERM_FAILURE);
      radius /= decrease_factor; decrease_factor *= 2.0;
      sum->num_unsuccessful_steps++;
      record(x_cost, 0, 0, 0, cg);
      continue;
    }
    num_invalid = 0;
    double cand_cost = h[SC_TRIAL];
    if (!std::isfinite(cand_cost)) { cand_cost = std::numeric_limits<double>::max(); sum->nonfinite = 1; }
    const double step_norm = std::sqrt(h[SC_STEP + 3]);
    const double cost_change = x_cost - cand_cost;
    const double rel_dec = cost_change / model_cost_change;
    if (forcing && loose_applied) {
      // A decision that hangs by less than a factor two -- stop here or go on, take the step or not -- is only the reference's if the POINT it is
      // taken at is the reference's: under the MAGSAC losses a cost change is a sum of table-cell jumps, and a state 1e-9 rad off moves it by
      // ten per cent (fuzz_forcing seed 9 trial 87: 0.9e-6 against 1.1e-6 of the cost at the function tolerance of 1e-6 -- 7 LM iterations
      // against 14, the same rotations).  The run has taken inexact steps, so it is redone with exact ones.
      // (Only where the decision moves the ANSWER: a candidate that lowers the cost -- stopping leaves it unapplied, going on applies it.  A
      // rejected candidate changes nothing whichever iteration the run ends at.  And a factor two only where the cost is a staircase; under a
      // smooth loss the decisive quantities of the two trajectories agree to ~1e-5 relative, the band is one per mille.)
      const double pt = o.parameter_tolerance * (x_norm + o.parameter_tolerance), ft = o.function_tolerance * x_cost, acc = std::fabs(cost_change);
      // (the staircase's noise in a cost change falls with the number of edges that make it up: 0.9 against 1.1 at 44k edges, 1.797 against
      // 1.815 at 300k -- the band is +- 100 / sqrt(E) relative, at most the factor two)
      // (the GLOBAL edge count, bit-identical on every rank of a sharded problem -- summed in the create-time agreement: with a rank-local count
      // the band would differ from rank to rank, and a decision inside the sliver between two ranks' bands would send one rank into the restart
      // while the others enter the next collective)
      const double n_e = std::fmax(1.0, P->sharded ? P->cost_n_global : (double)P->cost.n);
      const double w = P->loss_staircase ? std::fmin(1.0, 100.0 / std::sqrt(n_e)) : 1e-3, lo = 1.0 / (1.0 + w), hi = 1.0 + w;
      if (cost_change > 0.0 && ((step_norm > lo * pt && step_norm <= hi * pt) || (acc > lo * ft && acc <= hi * ft) || (rel_dec > lo * o.min_relative_decrease && rel_dec <= hi * o.min_relative_decrease))) {
        record(x_cost, cost_change, step_norm, rel_dec, cg); finish(GSFM_TERM_NO_CONVERGENCE); return GSFM_INTERNAL_RESTART;
      }
    }
    if (sum->iters_to_1e6 < 0 && std::fabs(cost_change) <= 1e-6 * x_cost) sum->iters_to_1e6 = iteration;
    if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { record(x_cost, cost_change, step_norm, rel_dec, cg); return finish(GSFM_TERM_PARAMETER_TOLERANCE); }
    if (std::fabs(cost_change) <= o.function_tolerance * x_cost) { record(x_cost, cost_change, step_norm, rel_dec, cg); return finish(GSFM_TERM_FUNCTION_TOLERANCE); }
    if (rel_dec > o.min_relative_decrease) {  // HandleSuccessfulStep
      std::swap(P->x.p, P->x_trial.p);
      // (a copy, not a pointer swap: the captured PCG / Cholesky graphs hold the address of the quaternions they rotate with)
      HIPCHK(hipMemcpyAsync(P->q.p, P->q_trial.p, 32 * (size_t)P->n_cams, hipMemcpyDeviceToDevice, P->stream));
      x_norm = std::sqrt(h[SC_STEP + 4]);
      x_cost = cand_cost;  // Ceres re-evaluates at the accepted point: same value
      if (int st = launch_lin(P, P->q.p)) return st;
      sum->num_residual_sweeps++; sum->num_linearizations++;
      radius = radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel_dec - 1.0, 3));   // (std::pow as Ceres' LevenbergMarquardtStrategy::StepAccepted and the oracle; k_lm_decide: the same value through lm_cube)
      radius = std::fmin(o.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      launch_prep(P, o, radius, false);
      prep_valid = true;
      if (defer_gmax) {
        HIPCHK(hipMemcpyAsync((void*)gmax_pin, P->scal.p + SC_GMAX, sizeof(double), hipMemcpyDeviceToHost, P->stream));
        gmax_deferred = true;
        gmax_trace_slot = P->trace.size() + 3;   // (column 3 of the row record() appends below)
      } else {
        if (int st = read_scalars(P, h)) return st;
        gmax = h[SC_GMAX];
      }
      sum->num_successful_steps++;
      last_successful = true;
      pred_rms = step_norm / sqrt_n;
      if (step_loose) loose_applied = true;
      if (forcing_live && prev_accepted_norm > 0.0 && step_norm > contraction_max * prev_accepted_norm) {   // the contraction gate (see above)
        forcing_live = false;
        if (loose_applied) { record(x_cost, cost_change, step_norm, rel_dec, cg); finish(GSFM_TERM_NO_CONVERGENCE); return GSFM_INTERNAL_RESTART; }
      }
      prev_accepted_norm = step_norm;
    } else {  // HandleUnsuccessfulStep
      radius /= decrease_factor; decrease_factor *= 2.0;
      sum->num_unsuccessful_steps++;
    }
    record(x_cost, cost_change, step_norm, rel_dec, cg);
  }
}

}  // namespace
