// Host side, part 3: the linear solve of an LM step by block-Jacobi PCG -- textbook recurrence (run_pcg), single-reduction recurrence
// (run_pcg2), the two-level preconditioner's coarse matrix (coarse_build) -- with the chunks between two host looks replayed as hipGraphs.
#pragma once
#include "host_common.hpp"

namespace {

// Inverse of a symmetric positive definite n x n matrix (row-major) through its Cholesky factor; false if a pivot is not positive.
bool spd_inverse(std::vector<double>& A, size_t n, std::vector<double>& inv) {
  for (size_t j = 0; j < n; ++j) {          // A <- L (lower), column by column
    double d = A[j * n + j];
    for (size_t k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double l = std::sqrt(d);
    A[j * n + j] = l;
    for (size_t i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (size_t k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / l;
    }
  }
  std::vector<double> Li(n * n, 0.0);       // L^-1 (lower), row by row: row_i = (e_i - sum_{k<i} L[i][k] row_k) / L[i][i]  (contiguous rows only)
  for (size_t i = 0; i < n; ++i) {
    double* ri = &Li[i * n];
    ri[i] = 1.0;
    for (size_t k = 0; k < i; ++k) {
      const double l = A[i * n + k];
      const double* rk = &Li[k * n];
      for (size_t c = 0; c <= k; ++c) ri[c] -= l * rk[c];
    }
    const double d = 1.0 / A[i * n + i];
    for (size_t c = 0; c <= i; ++c) ri[c] *= d;
  }
  inv.assign(n * n, 0.0);                   // A^-1 = L^-T L^-1 as a sum of rank-one updates of the lower triangle, then mirrored
  for (size_t k = 0; k < n; ++k) {
    const double* rk = &Li[k * n];
    for (size_t i = 0; i <= k; ++i) {
      const double a = rk[i];
      double* oi = &inv[i * n];
      for (size_t j = 0; j <= i; ++j) oi[j] += a * rk[j];
    }
  }
  for (size_t i = 0; i < n; ++i) for (size_t j = 0; j < i; ++j) inv[j * n + i] = inv[i * n + j];
  return true;
}

// Coarse matrix of the two-level preconditioner for the current linearisation and damping: assembled on the device, inverted on the host
// (3 n_agg <= 384 unknowns).  Leaves P->coarse_n = 0 (plain block-Jacobi for this step) if the matrix is not positive definite.
int coarse_build(gsfm_rot_problem* P, bool pcg_struggles) {
  P->coarse_n = 0;
  if (!P->coarse_want || !P->lin_is_lap || (P->coarse_adaptive && !pcg_struggles)) return 0;
  const uint32_t na = P->coarse_want, nc = 3 * na;   // (buffers: allocated at creation, before the ranks of a sharded problem vote)
  const int tk = P->timer.begin(T_CG);
  HIPCHK(hipMemsetAsync(P->coarseA.p, 0, 8 * (size_t)nc * nc, P->stream));
  CoarseAsmArgs a{};
  a.n_rows = P->n_rows; a.G = P->G; a.n_agg = na; a.chunk = P->coarse_chunk; a.row_ptr = P->row_ptr.p; a.col = P->col.p;
  a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.Mblk = P->Mblk.p; a.q = P->q_lin; a.Ac = P->coarseA.p;
  a.row_base = P->own_begin; a.scale = P->coarse_scale.p;
  {  // the fixed-point scale must be the same on every rank: the largest diagonal entry over ALL cameras (Mblk is complete everywhere)
    const uint32_t nb = (uint32_t)grid_for(P->n_cams);
    hipLaunchKernelGGL(k_coarse_scale, dim3(nb), dim3(GSFM_BLOCK), 0, P->stream, (const double*)P->Mblk.p, P->n_cams, 0u, P->part_a.p, P->coarse_scale.p, 0);
    hipLaunchKernelGGL(k_coarse_scale, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, (const double*)P->Mblk.p, nb, 0u, P->part_a.p, P->coarse_scale.p, 1);
  }
  if (P->n_rows) hipLaunchKernelGGL(k_coarse_assemble, dim3(grid_for((size_t)P->n_rows * P->G)), dim3(GSFM_BLOCK), 0, P->stream, a);
  hipLaunchKernelGGL(k_coarse_unscale, dim3(grid_for((size_t)nc * nc)), dim3(GSFM_BLOCK), 0, P->stream, P->coarseA.p, (size_t)nc * nc, (const double*)P->coarse_scale.p);
  P->timer.end(tk);
  // sharded: every rank summed the rows it owns; the vector side (restriction, coarse solve, prolongation) then runs replicated on the
  // replicated PCG vectors like every other O(N) step, so this all-reduce per LM step is the only collective the preconditioner adds
  if (int st = all_reduce(P, P->coarseA.p, (size_t)nc * nc)) return st;
  P->h_coarse.resize((size_t)nc * nc);
  HIPCHK(hipMemcpyAsync(P->h_coarse.data(), P->coarseA.p, 8 * (size_t)nc * nc, hipMemcpyDeviceToHost, P->stream));
  const double t_a = now_ms();
  if (int st = sync_check(P, "coarse matrix")) return st;
  const double t_b = now_ms();
  auto& A = P->h_coarse;
  for (size_t i = 0; i < nc; ++i) for (size_t j = 0; j < i; ++j) { const double v = 0.5 * (A[i * nc + j] + A[j * nc + i]); A[i * nc + j] = A[j * nc + i] = v; }
  for (size_t i = 0; i < nc; ++i) if (A[i * nc + i] == 0.0) A[i * nc + i] = 1.0;   // an aggregate of cameras without edges: decoupled, its correction stays zero
  if (!spd_inverse(A, nc, P->h_coarse_inv)) return 0;
  if (getenv("GSFM_COARSE_TIMING")) fprintf(stderr, "gsfm coarse: assemble + download (wait) %.2f ms, host inverse of %u unknowns %.2f ms\n", t_b - t_a, nc, now_ms() - t_b);
  HIPCHK(hipMemcpyAsync(P->coarseAinv.p, P->h_coarse_inv.data(), 8 * (size_t)nc * nc, hipMemcpyHostToDevice, P->stream));
  P->coarse_n = na;
  return 0;
}

// May a chunk of PCG iterations of a SHARDED problem be captured into a hipGraph together with its collectives?  Only if the
// communicator's callbacks do nothing but enqueue work on the solver's stream (GSFM_SHARD_CAPTURABLE: the native RCCL communicator).
// pcg_hip_graph = 1 (default) then captures; 2 is the old explicit opt-in and means the same; GSFM_PCG_GRAPH_COLLECTIVES=0 switches the
// capture of collectives off (plain launches), e.g. to isolate a communicator problem.
bool graph_collectives_ok(const gsfm_rot_problem* P, const gsfm_rot_options& o) {
  if (!(P->shard.flags & GSFM_SHARD_CAPTURABLE) || o.pcg_hip_graph < 1) return false;
  const char* e = getenv("GSFM_PCG_GRAPH_COLLECTIVES");
  return !(e && *e && atoi(e) == 0);
}

// ---- the PCG's status without a stream synchronisation (kernels.hpp, k_pcg_mail) -----------------------------------------------------------
// mail_usable: allocates on first use (mapped, coherent host memory + the device counter); the read-backs remain the fallback when that fails.
bool mail_usable(gsfm_rot_problem* P) {
  if (P->mail_state == 0) {
    P->mail_state = -1;
    if (hipHostMalloc((void**)&P->mail_host, (GSFM_MAIL_WORDS + 1) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      std::memset(P->mail_host, 0, (GSFM_MAIL_WORDS + 1) * sizeof(double));
      if (hipHostGetDevicePointer((void**)&P->mail_dev, P->mail_host, 0) == hipSuccess && P->mail_count.alloc(1, true) == hipSuccess) { P->mail_expected = 0.0; P->mail_state = 1; }
    }
    if (P->mail_state < 0) { (void)hipGetLastError(); if (P->mail_host) { (void)hipHostFree(P->mail_host); P->mail_host = nullptr; } }
  }
  return P->mail_state > 0;
}
// enqueue (or capture) the post of `bytes` of the scalar block at sc
void mail_post(gsfm_rot_problem* P, const void* sc, size_t bytes) {
  static_assert(sizeof(CgScalars) % 8 == 0 && sizeof(Cg2Scalars) % 8 == 0 && sizeof(CgScalars) <= 8 * GSFM_MAIL_WORDS && sizeof(Cg2Scalars) <= 8 * GSFM_MAIL_WORDS, "scalar blocks fit the mailbox");
  hipLaunchKernelGGL(k_pcg_mail, dim3(1), dim3(1), 0, P->stream, (const double*)sc, (int)(bytes / 8), P->mail_dev, P->mail_count.p);
}
// wait for post number mail_expected and take `bytes` of it.  The wait is bounded: after 20 s the stream is synchronised, which surfaces a device
// error if there was one.
int mail_wait(gsfm_rot_problem* P, void* dst, size_t bytes) {
  volatile double* stamp = P->mail_host + GSFM_MAIL_WORDS;
  const double t0 = now_ms();
  bool seen = false;
  for (long spin = 0; !(seen = *stamp == P->mail_expected); ++spin) {
    if ((spin & 0x3ff) == 0x3ff && now_ms() - t0 > 20000.0) break;
    __builtin_ia32_pause();
  }
  if (!seen) {
    if (int st = sync_check(P, "pcg status")) return st;
    if (*stamp != P->mail_expected) return fail(GSFM_ERR_HIP, "the PCG status of a chunk never arrived");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  std::memcpy(dst, (const void*)P->mail_host, bytes);
  return 0;
}

// A struggling solve: past its first 1000 iterations PCG goes on only while the relative residual still halves within 512 iterations (looked at
// when the host looks: the scalars are replicated, every rank of a sharded problem decides alike).  A solve that stops here -- or at
// max_cg_iterations -- above its tolerance is reported by lm_solve (num_pcg_capped_steps), never taken silently.
struct PcgStagnation {
  double mark_rel = 1.0; int mark_it = 0;
  bool stop(int iters, double rel) {
    if (rel < 0.5 * mark_rel) { mark_rel = rel; mark_it = iters; return false; }
    return iters >= 1000 && iters - mark_it >= 512;
  }
};

// floor^2 of the absolute tolerance (kernels.hpp, k_cam_bound): 2e-14 rad -- below what a relative residual of 1e-12 leaves on a step of a degree
// For a DISCONNECTED problem under a smooth loss the floor is 1e-11 rad: its scenes are independent problems that converge at their own pace,
// and the ones that have converged must stop costing iterations while the slowest iterates on (C4: 13 of 14); what they are left short of is
// their conditioning times 1e-11 rad, five orders inside the bar.  (Not under MAGSAC: an iterate 1e-11 rad off can sit in another table cell.)
// (gsfm_rot_options::component_rest = 0 switches both departures from the reference's single global stopping rule off: P->component_rest, set by lm_solve)
double pcg_abs_floor2(const gsfm_rot_problem* P) { const double f = (P->n_components > 1 && !P->loss_staircase && !P->cb && P->component_rest) ? 1e-11 : 2e-14; return f * f; }
// ... and a factorised component whose exact step has fallen below this is put to rest for the remainder of the solve (comp_kernels.hpp, k_comp_activity)
double comp_freeze_below(const gsfm_rot_problem* P) { return (!P->loss_staircase && !P->cb && P->component_rest) ? 1e-10 : 0.0; }

// block-Jacobi PCG on (J^T J + Lambda) eta = -g to the relative residual `tol` -- or, etol2 > 0 (a loose solve of the forcing schedule), until the
// estimated relative energy-norm error squared falls below etol2 (kernels.hpp, cg_energy_stop); returns iterations.  resume_iters >= 0: continue the solve
// that stopped after that many iterations (at a looser tolerance) instead of starting one -- the device state is exactly what the stopping
// iteration left (kernels.hpp, CgScalars::done_seen), so the iterates are those of an uninterrupted solve at `tol`.
int run_pcg(gsfm_rot_problem* P, const gsfm_rot_options& o, double tol, double etol2, int resume_iters, int* iters_out, double* rel_out) {
  const bool resume = resume_iters >= 0;
  // PACKED sharded problem: this rank's own block of a block-diagonal system, solved without a collective -- right-hand side zero outside the
  // rank's cameras (everything the recurrence touches stays zero there), mat-vec without the gather; the step and the residual are gathered
  // once per evaluated step by the caller (solver_components.hpp, packed_exchange).  The ranks' solves end after different iteration counts: nothing in the LM loop is decided from them.
  struct LocalScope { gsfm_rot_problem* P; bool on; ~LocalScope() { if (on) { P->pcg_local = false; P->b_rhs = nullptr; } } } local{P, P->sharded && P->packed};
  if (local.on) {
    if (!P->b_own.p && P->b_own.alloc(3 * (size_t)P->n_cams, true) != hipSuccess) return fail(GSFM_ERR_HIP, "allocating the rank-local right-hand side failed");
    if (!resume) hipLaunchKernelGGL(k_mask_range, dim3(grid_for(P->n_cams)), dim3(GSFM_BLOCK), 0, P->stream, (const double*)(P->b_rhs ? P->b_rhs : P->b.p), P->own_begin, P->own_end, P->n_cams, P->b_own.p);
    P->b_rhs = P->b_own.p; P->pcg_local = true;
  }
  CgArgs a{};
  a.n = P->n_cams; a.nb = P->nb_cam; a.par = 0; a.tol = tol; a.etol2 = etol2; a.max_iters = o.max_cg_iterations; a.stall_limit = o.cg_stall_iterations;
  a.Minv = P->Minv.p; a.b = P->b_rhs ? P->b_rhs : P->b.p; a.xcg = P->xcg.p; a.r = P->r.p; a.z = P->z.p; a.p = P->p.p; a.Ap = P->Ap.p;
  a.part_a = P->part_a.p; a.part_b = P->part_b.p; a.sc = P->cgsc.p;
  a.zbound = P->scal.p + SC_ZBOUND; a.abs_floor2 = pcg_abs_floor2(P);
  a.q = P->q_lin; a.u = P->lin_is_lap ? P->u_rot.p : nullptr;
  a.coarse_n = P->coarse_n; a.coarse_chunk = P->coarse_chunk; a.xc = P->coarse_xc.p; a.active = P->active.p;
  // aggregates at least as wide as a block of the camera kernels (always, unless forced narrower): the restriction rides along in k_cg_update
  const bool fused_restrict = P->coarse_n && P->coarse_chunk >= GSFM_BLOCK;
  a.rc_part = fused_restrict ? P->coarse_part.p : nullptr;
  CoarseArgs ca{};
  ca.n = P->n_cams; ca.n_agg = P->coarse_n; ca.chunk = P->coarse_chunk; ca.q = P->q_lin; ca.r = P->r.p; ca.rc = P->coarse_rc.p; ca.Ainv = P->coarseAinv.p;
  ca.xc = P->coarse_xc.p; ca.done = nullptr; ca.active = P->active.p; ca.rc_part = nullptr; ca.nb = (uint32_t)P->nb_cam;
  const dim3 g(P->nb_cam), blk(GSFM_BLOCK);
  const int tk0 = P->timer.begin(T_CG);
  if (resume) hipLaunchKernelGGL(k_cg_resume, dim3(1), dim3(1), 0, P->stream, P->cgsc.p, tol, etol2, a.max_iters);
  else {
  hipLaunchKernelGGL(k_cg_init, g, blk, 0, P->stream, a);
  hipLaunchKernelGGL(k_cg_init_fin, dim3(1), blk, 0, P->stream, a);
  if (a.coarse_n) {   // z_0 = Minv r_0 + P Ac^-1 P^T r_0
    hipLaunchKernelGGL(k_coarse_restrict, dim3(a.coarse_n), blk, 0, P->stream, ca);
    hipLaunchKernelGGL(k_coarse_apply, dim3(1), dim3(1024), 0, P->stream, ca);
    hipLaunchKernelGGL(k_cg_init_coarse, g, blk, 0, P->stream, a);
    hipLaunchKernelGGL(k_cg_init_coarse_fin, dim3(1), dim3(1), 0, P->stream, a);
  }
  }
  ca.done = &P->cgsc.p->done; ca.rc_part = a.rc_part;
  P->timer.end(tk0);
  CgScalars h{};
  const int chunk = std::max(1, o.cg_check_interval);
  auto enqueue_iter = [&]() -> int {
    bool dotted = false;
    if (int st = launch_matvec(P, P->Mblk.p, P->p.p, P->Ap.p, &P->cgsc.p->done, a.part_a, &dotted)) return st;
    if (P->sharded && !P->pcg_local) P->n_pcg_collectives++;
    if (!dotted) hipLaunchKernelGGL(k_cg_dot, g, blk, 0, P->stream, a);
    hipLaunchKernelGGL(k_cg_update, g, blk, 0, P->stream, a);
    if (a.coarse_n) {
      if (!fused_restrict) hipLaunchKernelGGL(k_coarse_restrict, dim3(a.coarse_n), blk, 0, P->stream, ca);
      hipLaunchKernelGGL(k_coarse_apply, dim3(1), dim3(1024), 0, P->stream, ca);
    }
    hipLaunchKernelGGL(k_cg_pupdate, g, blk, 0, P->stream, a);
    a.par ^= 1;
    return 0;
  };
  const bool mail = mail_usable(P);
  auto enqueue_chunk = [&]() -> int {  // `chunk` iterations (+ the status post); leaves a.par where it found it when chunk is even
    for (int c = 0; c < chunk; ++c) if (int st = enqueue_iter()) return st;
    if (mail) mail_post(P, P->cgsc.p, sizeof(CgScalars));
    return 0;
  };
  // The chunk between two host checks as one hipGraph launch: 4 * chunk dependent kernels whose arguments never change.
  auto& G = P->pcg_graph;
  bool graph = o.pcg_hip_graph && (!P->sharded || P->pcg_local || graph_collectives_ok(P, o)) && chunk % 2 == 0 && !G.unusable;
  // (the tolerance is device-resident, CgScalars::tol: a captured chunk serves every tolerance)
  if (graph && (!G.exec || G.max_iters != a.max_iters || G.stall != a.stall_limit || G.chunk != chunk || G.lap != P->lin_is_lap || G.coarse != a.coarse_n)) {
    G.reset();
    hipGraph_t captured = nullptr;
    if (hipStreamBeginCapture(P->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      const int c0 = P->n_collectives, p0 = P->n_pcg_collectives;
      const int st = enqueue_chunk();
      const hipError_t e = hipStreamEndCapture(P->stream, &captured);
      G.collectives = P->n_collectives - c0;            // captured, not issued: counted per replay below
      P->n_collectives = c0; P->n_pcg_collectives = p0;
      if (st == 0 && e == hipSuccess && captured && hipGraphInstantiate(&G.exec, captured, nullptr, nullptr, 0) == hipSuccess) {
        G.tol = a.tol; G.max_iters = a.max_iters; G.stall = a.stall_limit; G.chunk = chunk; G.lap = P->lin_is_lap; G.coarse = a.coarse_n;
      } else { G.exec = nullptr; }
      if (captured) (void)hipGraphDestroy(captured);
    }
    if (!G.exec) { (void)hipGetLastError(); G.unusable = true; graph = false; }  // e.g. a stream that cannot be captured: plain launches
  }
  int launched = 0, chunks = 1;
  PcgStagnation stagnation;
  if (resume) {   // the parity of the iteration that follows the stop; a captured chunk starts at parity 0
    launched = resume_iters;
    if (resume_iters & 1) {
      a.par = 1;
      const int tk = P->timer.begin(T_CG);
      if (int st = enqueue_iter()) return st;
      P->timer.end(tk);
      ++launched; P->n_pcg_launched++;
    }
  }
  while (true) {
    const int tk = P->timer.begin(T_CG);
    for (int c = 0; c < chunks; ++c) {
      if (graph) { HIPCHK(hipGraphLaunch(G.exec, P->stream)); P->graph_launches++; P->n_collectives += G.collectives; P->n_pcg_collectives += G.collectives; }
      else if (int st = enqueue_chunk()) return st;
      launched += chunk; P->n_pcg_launched += chunk;
      if (mail) P->mail_expected += 1.0;
    }
    P->timer.end(tk);
    if (int st = mail ? mail_wait(P, &h, sizeof(h)) : read_back(P, &h, P->cgsc.p, sizeof(h), "pcg")) return st;
    if (h.done || launched >= o.max_cg_iterations + chunk || stagnation.stop(h.iters, h.last_rel)) break;
    // Fewer host round trips: extrapolate the average convergence factor so far to the tolerance and enqueue that many
    // chunks before looking again (kernels past convergence return at their first instruction, so overshoot is cheap).
    chunks = 1;
    if (!(etol2 > 0.0) && h.iters > 0 && h.last_rel > 0.0 && h.last_rel < 1.0 && tol > 0.0 && tol < h.last_rel) {
      const double per_iter = std::log(h.last_rel) / h.iters;
      const double remaining = std::log(tol / h.last_rel) / per_iter;
      chunks = (int)std::min(8.0, std::max(1.0, std::ceil(remaining / chunk)));
    }
    chunks = std::min(chunks, std::max(1, (o.max_cg_iterations + chunk - launched + chunk - 1) / chunk));
  }
  *iters_out = h.iters; *rel_out = h.last_rel <= h.tol ? std::fmin(h.last_rel, tol) : h.last_rel;   // (converged against the floor-adjusted tolerance: converged)
  return 0;   // (packed: the step and the residual are gathered by the caller, once per evaluated step: packed_exchange)
}

// single-reduction PCG (Chronopoulos-Gear): 2 kernels per iteration (3 + one all-gather when sharded); the launch-latency regime's
// default (see use_single_reduction).  The chunk between two host checks replays as one hipGraph, like run_pcg's.
int run_pcg2(gsfm_rot_problem* P, const gsfm_rot_options& o, double tol, double etol2, int resume_iters, int* iters_out, double* rel_out) {
  const bool resume = resume_iters >= 0;
  Cg2Args c{};
  const int nb_mv = P->nb_mv, reps = P->mv_reps;
  c.n = P->n_cams; c.nb_cam = P->nb_cam; c.n_part_d = (P->sharded || P->cs.active) ? P->nb_cam : nb_mv; c.par = 0; c.first = 1; c.tol = tol; c.etol2 = etol2; c.max_iters = o.max_cg_iterations;
  c.Minv = P->Minv.p; c.b = P->b_rhs ? P->b_rhs : P->b.p; c.x = P->xcg.p; c.r = P->r.p; c.u = P->z.p; c.w = P->Ap.p; c.p = P->p.p; c.s = P->s_dir.p;
  c.part_g = P->part_g2.p; c.part_d = P->part_d2.p; c.sc = P->cg2sc.p;
  c.zbound = P->scal.p + SC_ZBOUND; c.abs_floor2 = pcg_abs_floor2(P);
  c.q = P->q_lin; c.urot = P->lin_is_lap ? P->u_rot.p : nullptr;
  // Sharded: A u and the delta partials of a rank's rows leave in ONE all-gather (slot = slice of w, then the partials); the mat-vec kernels
  // address y by global camera index, so they get the slot's base shifted back by the rank's first camera.
  double* w_own = P->Ap.p;          // what the mat-vec writes through (indexed 3 * global camera)
  double* dots_own = P->part_d2.p;  // where its delta partials go
  if (P->sharded) {
    const uint32_t slice = P->shard.slice_width, tail = P->w_tail, stride = 3 * slice + tail;
    if (!P->w_gather.p) return fail(GSFM_ERR_HIP, "the all-gather buffer of the sharded PCG was not allocated");   // (create allocates it, before the ranks agree)
    c.w = P->w_gather.p; c.w_stride = stride; c.w_slice = slice; c.w_tail = tail; c.n_part_d = (int)(tail * P->shard.world_size);
    double* slot = P->w_gather.p + (size_t)P->shard.rank * stride;
    w_own = slot - 3 * (size_t)P->own_begin; dots_own = slot + 3 * (size_t)slice;
  }
  MatvecCgArgs m{};
  m.mv.n_rows = P->n_rows; m.mv.row_base = P->own_begin; m.mv.G = P->G; m.mv.row_ptr = P->row_ptr.p; m.mv.col = P->col.p;
  m.mv.h0 = P->h0.p; m.mv.h1 = P->h1.p; m.mv.h2 = P->h2.p; m.mv.h3 = P->h3.p; m.mv.h4 = P->h4.p; m.mv.Mblk = P->Mblk.p;
  m.mv.p = P->z.p; m.mv.y = w_own; m.mv.done = nullptr; m.mv.q = P->q_lin; m.mv.u = P->u_rot.p; m.with_dots = 1; m.reps = (uint32_t)reps;
  const dim3 gcam(P->nb_cam), gmv(nb_mv), blk(GSFM_BLOCK);
  const int tk0 = P->timer.begin(T_CG);
  if (resume) hipLaunchKernelGGL(k_cg2_resume, dim3(1), dim3(1), 0, P->stream, P->cg2sc.p, tol, etol2);
  else hipLaunchKernelGGL(k_cg2_init, gcam, blk, 0, P->stream, c);
  P->timer.end(tk0);
  Cg2Scalars h{};
  const int chunk = std::max(1, o.cg_check_interval);
  // One iteration = mat-vec (+ delta partials), vector step.  `first` / `par` are by-value kernel arguments: a captured chunk must
  // start at par == 0, first == 0, so the very first iteration is launched plainly and chunks have even length.
  auto enqueue_iter = [&]() -> int {
    m.cg = c; m.cg.part_d = dots_own;
    if (P->cs.active) {   // column-sorted layout: K3c with the same entry decision, delta partials from its finishing kernel (one per camera block)
      auto& L = P->cs;
      ColMatvecCgArgs cm{};
      cm.mv.L = L.dev(); cm.mv.b0 = P->h0.p; cm.mv.b1 = P->h1.p; cm.mv.b2 = P->h2.p; cm.mv.u = P->u_rot.p; cm.mv.part = L.part.p; cm.mv.done = nullptr; cm.cg = c;
      if (L.k16_active) hipLaunchKernelGGL(k_mv_col_cg<true>, dim3(L.n_wg), dim3(GSFM_K3C_THREADS), 0, P->stream, cm);
      else hipLaunchKernelGGL(k_mv_col_cg<false>, dim3(L.n_wg), dim3(GSFM_K3C_THREADS), 0, P->stream, cm);
      ColFinishArgs f{};
      f.n_rows = P->n_rows; f.row_base = P->own_begin; f.nch = L.nch; f.n_wg = L.n_wg; f.part = L.part.p; f.Mblk = P->Mblk.p; f.p = P->z.p; f.q = P->q_lin; f.y = w_own;
      f.done = &P->cg2sc.p->done; f.dot_part = dots_own;
      hipLaunchKernelGGL(k_mv_col_finish, dim3(grid_for(P->n_rows)), dim3(GSFM_BLOCK), 0, P->stream, f);
    }
    else if (P->lin_is_lap) hipLaunchKernelGGL(k_matvec_cg<true>, gmv, blk, 0, P->stream, m);
    else hipLaunchKernelGGL(k_matvec_cg<false>, gmv, blk, 0, P->stream, m);
    if (P->sharded) {
      if (int st = all_gather(P, P->w_gather.p, (size_t)c.w_stride)) return st;
      P->n_pcg_collectives++;
    }
    hipLaunchKernelGGL(k_cg2_step, gcam, blk, 0, P->stream, c);
    c.par ^= 1; c.first = 0;
    return 0;
  };
  int launched = 0;
  if (resume) {   // the launch that stopped had parity resume_iters & 1 (and `first` set only if nothing had run): take it again, plainly, up to parity 0
    launched = resume_iters; c.par = resume_iters & 1; c.first = resume_iters == 0;
    const int tk = P->timer.begin(T_CG);
    while (c.par || c.first) { if (int st = enqueue_iter()) return st; ++launched; P->n_pcg_launched++; }
    P->timer.end(tk);
  } else {  // iterations 0 and 1 (first = 1, then par = 1): plain launches; afterwards par == 0 at every chunk start
    const int tk = P->timer.begin(T_CG);
    for (int k = 0; k < 2; ++k) { if (int st = enqueue_iter()) return st; ++launched; P->n_pcg_launched++; }
    P->timer.end(tk);
  }
  const bool mail = mail_usable(P);
  auto& G = P->pcg2_graph;
  bool graph = o.pcg_hip_graph && (!P->sharded || graph_collectives_ok(P, o)) && chunk % 2 == 0 && !G.unusable && !P->pcg_graph.unusable;
  if (graph && (!G.exec || G.max_iters != c.max_iters || G.chunk != chunk || G.lap != P->lin_is_lap)) {
    G.reset();
    hipGraph_t captured = nullptr;
    if (hipStreamBeginCapture(P->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      int st = 0;
      const int c0 = P->n_collectives, p0 = P->n_pcg_collectives;
      for (int k = 0; k < chunk && st == 0; ++k) st = enqueue_iter();
      if (mail && st == 0) mail_post(P, P->cg2sc.p, sizeof(Cg2Scalars));
      const hipError_t e = hipStreamEndCapture(P->stream, &captured);
      G.collectives = P->n_collectives - c0;
      P->n_collectives = c0; P->n_pcg_collectives = p0;
      if (st == 0 && e == hipSuccess && captured && hipGraphInstantiate(&G.exec, captured, nullptr, nullptr, 0) == hipSuccess) {
        G.tol = c.tol; G.max_iters = c.max_iters; G.chunk = chunk; G.lap = P->lin_is_lap;
      } else { G.exec = nullptr; }
      if (captured) (void)hipGraphDestroy(captured);
    }
    if (!G.exec) { (void)hipGetLastError(); G.unusable = true; graph = false; }
  }
  int chunks = 1;
  PcgStagnation stagnation;
  while (true) {
    const int tk = P->timer.begin(T_CG);
    for (int cc = 0; cc < chunks; ++cc) {
      if (graph) { HIPCHK(hipGraphLaunch(G.exec, P->stream)); P->graph_launches++; P->n_collectives += G.collectives; P->n_pcg_collectives += G.collectives; }
      else { for (int k = 0; k < chunk; ++k) if (int st = enqueue_iter()) return st; if (mail) mail_post(P, P->cg2sc.p, sizeof(Cg2Scalars)); }
      launched += chunk; P->n_pcg_launched += chunk;
      if (mail) P->mail_expected += 1.0;
    }
    P->timer.end(tk);
    if (int st = mail ? mail_wait(P, &h, sizeof(h)) : read_back(P, &h, P->cg2sc.p, sizeof(h), "pcg")) return st;
    if (h.done || launched >= o.max_cg_iterations + chunk + 2 || stagnation.stop(h.iters, h.last_rel)) break;
    chunks = 1;   // same look-ahead as run_pcg: extrapolate the convergence factor, enqueue that many chunks before looking again
    if (!(etol2 > 0.0) && h.iters > 0 && h.last_rel > 0.0 && h.last_rel < 1.0 && tol > 0.0 && tol < h.last_rel) {
      const double per_iter = std::log(h.last_rel) / h.iters;
      const double remaining = std::log(tol / h.last_rel) / per_iter;
      chunks = (int)std::min(8.0, std::max(1.0, std::ceil(remaining / chunk)));
    }
  }
  *iters_out = h.iters; *rel_out = h.last_rel <= h.tol ? std::fmin(h.last_rel, tol) : h.last_rel;   // (converged against the floor-adjusted tolerance: converged)
  return 0;
}

// Which PCG: the single-reduction variant halves the dependent launches per iteration (2 instead of 4), which is what bounds small
// graphs (tools/small_graph_pcg.py: 22 -> 14 -> 8 us per iteration at C2 size); from ~1M directed entries on the kernels dominate
// and the textbook recurrence is kept (its residual is the recursively updated one of the reference description).
bool use_single_reduction(const gsfm_rot_problem* P, const gsfm_rot_options& o) {
  // A PACKED sharded problem runs every rank's own PCG without a collective (run_pcg's LocalScope); run_pcg2 has no such scope -- its all-gather
  // sits inside the loop and the ranks' iteration counts differ -- so not even an explicit pcg_single_reduction = 1 selects it there (round-5 advisor).
  if (P->packed) return false;
  if (o.pcg_single_reduction >= 0) return o.pcg_single_reduction != 0;
  if (o.cg_stall_iterations > 0) return false;   // stagnation detection lives in the textbook variant's scalar kernel
  // (On the column-sorted layout the variant exists too -- k_mv_col_cg, one vector kernel instead of two -- and measures the same as the
  // textbook recurrence at C5: 30.69 against 30.63 ms per solve, 129 iterations both; the entry decision of its mat-vec costs what the
  // saved launch gains.  tools/archive/r03_pcg_variants.py)
  // Sharded: always -- there every launch counts (the per-rank kernels shrink with the rank count, the launches do not), and the variant is
  // 3 kernels + 1 collective per iteration (mat-vec, finish, [all-gather of A u with the delta partials in its tail], vector step) against
  // 5 + 1 for the textbook recurrence; except at tolerances below 1e-13 (disconnected graphs, lm_solve), where the recursively updated
  // residual of the textbook form is the safer one.
  if (P->sharded) return o.cg_relative_tolerance >= 1e-13 && P->n_components <= 1;
  return P->dir.n <= (size_t)2000000;
}

}  // namespace
